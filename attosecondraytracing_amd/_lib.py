"""Loader of libart_hip.so -- the only compute backend of this package.

There is NO CPU fallback: if the shared library was not built (`python -c "import __graft_entry__ as g;
g.build()"`), or no gfx950 device is visible, the first attempt to trace rays raises RuntimeError.
"""
import ctypes as C
import os

import numpy as np
import torch

from . import _abi

_HERE = os.path.dirname(os.path.abspath(__file__))
# The product library, in-tree.  No environment variable redirects the loader: a diagnostic BUILD is loaded by handing its
# path to HipBackend(path=...) explicitly (tools/ab_kernel.py does), never by the process-wide get_backend().
LIB_PATH = os.path.join(_HERE, "libart_hip.so")

_BACKEND = None


class ArtError(RuntimeError):
    pass


class HipBackend:
    """Thin typed wrapper: torch tensors in, C-ABI calls out.  All calls are asynchronous on torch's
    current HIP stream of the bundle's device.  Stream-safe like the C ABI underneath: every piece of scratch memory and
    every staging pair the wrapper reuses between calls is kept PER STREAM (keyed by the current stream's handle), so two
    torch streams may issue read-outs, compactions, scene launches and analyses concurrently."""

    name = "hip"

    def __init__(self, path=LIB_PATH):
        if not os.path.exists(path):
            raise RuntimeError(
                f"{path} not found: the HIP extension has not been built. Run "
                "`python -c 'import __graft_entry__ as g; g.build()'` (needs hipcc). There is no CPU fallback.")
        self.lib = C.CDLL(path)
        self.fn = _abi.bind(self.lib)
        v = self.fn["art_abi_version"]()
        if v != _abi.ART_ABI_VERSION:
            raise RuntimeError(f"libart_hip.so ABI version {v} != expected {_abi.ART_ABI_VERSION}: rebuild it")
        if not torch.cuda.is_available():
            raise RuntimeError("No HIP device visible to PyTorch: ray tracing needs an MI355X (gfx950). "
                               "There is no CPU fallback.")
        n = self.fn["art_device_count"]()
        if n <= 0:
            raise RuntimeError("libart_hip.so found no gfx950 device (art_device_count() = %d: %s)"
                               % (n, self.last_error()))
        self.device = torch.device("cuda", torch.cuda.current_device())
        self._scratch = {}
        self._side_streams = {}
        self._scene_pool, self._scene_uploads = {}, {}
        self._job_pool = {}
        # bookkeeping for measurements: launches of the fused kernels over at least `count_from` slots, in issue order
        # (bench.py reports which of them lie inside its timed region, so that a kernel trace can be cut to it)
        self.count_from, self.counted_launches = None, 0
        self.trace_events = None   # set to a list to collect (start, end) HIP events around each trace launch
        self.readout_events = None  # likewise around each fused read-out (kernel + final fold)

    # ------------------------------------------------------------------ helpers
    def last_error(self):
        return self.fn["art_last_error"]().decode("utf-8", "replace")

    def check(self, rc, what):
        if rc != 0:
            raise ArtError(f"{what} failed with code {rc}: {self.last_error()}")

    def stream_ptr(self):
        return C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    def synchronize(self):
        torch.cuda.current_stream(self.device).synchronize()

    def empty(self, n, dtype=torch.float64):
        return torch.empty(int(n), dtype=dtype, device=self.device)

    def zeros(self, n, dtype=torch.float64):
        return torch.zeros(int(n), dtype=dtype, device=self.device)

    def from_numpy(self, a, dtype=None):
        t = torch.from_numpy(np.ascontiguousarray(a))
        if dtype is not None:
            t = t.to(dtype)
        return t.to(self.device)

    def stream_key(self):
        return torch.cuda.current_stream(self.device).cuda_stream

    def side_stream_context(self, name):
        """`with be.side_stream_context(name):` -- torch's current stream becomes a stream of this backend's own (one per
        name) that has NOT waited for the caller's stream: for small latency-critical work (the alignment rays of a
        placement) whose read-backs must not queue behind bulk kernels.  Tensors created inside belong to that stream."""
        st = self._side_streams.get(name)
        if st is None:
            st = self._side_streams[name] = torch.cuda.Stream(device=self.device)
        return torch.cuda.stream(st)

    def scratch(self, key, n, dtype):
        """Reused scratch memory `key` of the CURRENT stream (work of one stream is ordered, so one area per stream and
        purpose is enough; two streams never share one)."""
        key = (key, self.stream_key())
        t = self._scratch.get(key)
        if t is None or t.numel() < n or t.dtype != dtype:
            t = torch.empty(int(max(n, 1)), dtype=dtype, device=self.device)
            self._scratch[key] = t
        return t

    # ------------------------------------------------------------------ entry points
    def _timed(self, call, sink="trace_events"):
        """Run one launch; when the sink list is set, bracket it with HIP events recorded on the launch stream."""
        events = getattr(self, sink)
        if events is None:
            return call()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        rc = call()
        e1.record()
        events.append((e0, e1))
        return rc

    def trace_element(self, desc, view_in, view_out, n):
        sp = self.stream_ptr()
        self.check(self._timed(lambda: self.fn["art_trace_element"](C.byref(desc), C.byref(view_in),
                                                                    C.byref(view_out), n, sp)), "art_trace_element")

    def trace_chain(self, descs, view_in, views_out, n, readout=None):
        """art_trace_chain, or art_trace_chain_readout when `readout` (from new_chain_readout) is given."""
        m = len(descs)
        if m > 1 or readout is not None:       # (a one-element chain without read-out is the per-element kernel's)
            self.note_launches(n, -(-m // 8))
        darr = (_abi.ArtElementDesc * m)(*descs)
        varr = (_abi.ArtBundleView * m)(*views_out)
        sp = self.stream_ptr()
        if readout is None:
            self.check(self._timed(lambda: self.fn["art_trace_chain"](darr, m, C.byref(view_in), varr, n, sp)),
                       "art_trace_chain")
        else:
            ro = readout["struct"]
            self.check(self._timed(lambda: self.fn["art_trace_chain_readout"](darr, m, C.byref(view_in), varr,
                                                                              C.byref(ro), n, sp)),
                       "art_trace_chain_readout")

    MAX_FUSED_READOUT_RAYS = 1 << 28     # one launch (art_trace_chain_readout)

    def new_chain_readout(self, ddesc, w, n, centres=(0.0, 0.0, 0.0), store=True, scratch=None, lite=False, targets=None):
        """Outputs + descriptor of a read-out fused behind a chain launch (ArtChainReadout): returns a dict with the
        result tensors 'X', 'Y', 'opl' (None with store=False), 'stats_dev' and the ctypes 'struct'.  lite=True: only
        count, sum of paths, bounding box and path range are reduced (ArtChainReadout.lite).  targets=(X, Y, opl): the
        read-out writes into these caller-owned tensors (e.g. the sections of a survivor send buffer: zero-copy gather)."""
        X = Y = opl = None
        if targets is not None:
            X, Y, opl = targets
            assert all(t.numel() >= n and t.dtype == torch.float64 and t.is_contiguous() for t in (X, Y, opl))
            store = True
        elif store:
            X, Y, opl = self.empty(n), self.empty(n), self.empty(n)
        out = self.empty(24)
        if scratch is None:
            scratch = self.scratch("chain_ro", self.fn["art_chain_readout_scratch_doubles"](n), torch.float64)
        ro = _abi.ArtChainReadout()
        ro.det = ddesc
        ro.w = None if w is None else w.data_ptr()
        ro.cx, ro.cy, ro.co = (float(v) for v in centres)
        ro.X, ro.Y, ro.opl = (None, None, None) if not store else (X.data_ptr(), Y.data_ptr(), opl.data_ptr())
        ro.scratch, ro.out24 = scratch.data_ptr(), out.data_ptr()
        ro.lite = 1 if lite else 0
        ro.sums = 0
        return {"struct": ro, "X": X, "Y": Y, "opl": opl, "P3": None, "stats_dev": out, "_keep": (w, scratch), "lite": bool(lite)}

    def new_chain_sums(self, w, n, scratch=None):
        """Descriptor of the SUMS tail (ArtChainReadout.sums): the tracing launch forms pass (1) of the analysis for the
        chain's last bundle -- count, sum point, sum vector, sum w, sum path in 'sums_dev'[0..8] -- while the ray is still
        in registers.  Hand 'sums_dev' to the analysis of that bundle (analysis.analyse does, through the bundle)."""
        out = self.empty(24)
        if scratch is None:
            scratch = self.scratch("chain_ro", self.fn["art_chain_readout_scratch_doubles"](n), torch.float64)
        ro = _abi.ArtChainReadout()
        ro.w = None if w is None else w.data_ptr()
        ro.X = ro.Y = ro.opl = None
        ro.scratch, ro.out24 = scratch.data_ptr(), out.data_ptr()
        ro.lite, ro.sums = 0, 1
        return {"struct": ro, "sums_dev": out, "_keep": (w, scratch), "sums": True}

    def chain_readout_scratch(self, n, count):
        """`count` scratch areas for fused read-outs of `count` chains in one scene launch."""
        per = int(self.fn["art_chain_readout_scratch_doubles"](n))
        t = torch.empty(per * count, dtype=torch.float64, device=self.device)
        return [t[k * per:(k + 1) * per] for k in range(count)]

    # scene table (include/art_hip.h: art_scene_bytes / art_scene_pack / art_trace_scene): many chains, one launch
    def scene_alloc(self, n_chains, n_elems, transient=False):
        """(pinned host image, device image) uint8 tensors of art_scene_bytes(n_chains, n_elems).
        transient=True (one-shot launches, RayTracingCalculationMany): the pair comes from a per-size pool instead of
        being allocated -- pinning host memory costs milliseconds.  A pooled pair is reused by the next call of the
        same size: the device image is only read by launches that were enqueued before the next upload on the same
        stream, and the pinned image is re-packed only after its last upload has completed (scene_pack waits)."""
        nb = int(self.fn["art_scene_bytes"](n_chains, n_elems))
        if nb <= 0:
            raise ArtError("art_scene_bytes: bad chain or element count")
        if transient:
            pkey = (nb, self.stream_key())          # per stream: another stream's launch may still read its device image
            pair = self._scene_pool.get(pkey)
            if pair is None:
                pair = self._scene_pool[pkey] = (torch.empty(nb, dtype=torch.uint8, pin_memory=True),
                                                 torch.empty(nb, dtype=torch.uint8, device=self.device))
            ev = self._scene_uploads.pop(pair[0].data_ptr(), None)
            if ev is not None:
                ev.synchronize()
            return pair
        return (torch.empty(nb, dtype=torch.uint8, pin_memory=True),
                torch.empty(nb, dtype=torch.uint8, device=self.device))

    def scene_pack(self, descs, views_in, views_out, n_chains, n_elems, host_image, readouts=None):
        """Pack descriptors (flat, chain-major), views and optional per-chain fused read-outs (new_chain_readout
        dicts) into the host image; returns the scene flags."""
        darr = (_abi.ArtElementDesc * (n_chains * n_elems))(*descs)
        iarr = (_abi.ArtBundleView * n_chains)(*views_in)
        oarr = (_abi.ArtBundleView * (n_chains * n_elems))(*views_out)
        rarr = None if readouts is None else (_abi.ArtChainReadout * n_chains)(*[r["struct"] for r in readouts])
        rc = self.fn["art_scene_pack"](darr, n_chains, n_elems, iarr, oarr, rarr, host_image.data_ptr())
        if rc < 0:
            self.check(rc, "art_scene_pack")
        return rc

    def scene_upload(self, host_image, dev_image):
        """Host image -> device image on the current stream; returns an event the host waits for before it re-packs."""
        dev_image.copy_(host_image, non_blocking=True)
        ev = torch.cuda.Event()
        ev.record()
        self._scene_uploads[host_image.data_ptr()] = ev
        return ev

    def note_launches(self, n, k=1):
        if self.count_from is not None and n >= self.count_from:
            self.counted_launches += k

    def trace_scene(self, dev_image, host_image, n, segments=1):
        """ONE launch (per 8 elements) for every chain of a packed scene; counts and flags come from the host image's
        header (art_scene_pack), of which dev_image is the uploaded copy."""
        self.note_launches(n, segments)
        sp = self.stream_ptr()
        self.check(self._timed(lambda: self.fn["art_trace_scene"](dev_image.data_ptr(), host_image.data_ptr(), n, sp)),
                   "art_trace_scene")

    def pack_rays(self, points, vectors, path0, n, view):
        self.check(self.fn["art_pack_rays"](points.data_ptr(), vectors.data_ptr(),
                                            None if path0 is None else path0.data_ptr(), n, C.byref(view),
                                            self.stream_ptr()), "art_pack_rays")

    def transform_bundle(self, M, T, rotate_points, view_in, view_out, n):
        m = (C.c_double * 9)(*[float(v) for v in np.asarray(M).reshape(9)])
        t = (C.c_double * 3)(*[float(v) for v in np.asarray(T).reshape(3)])
        self.check(self.fn["art_transform_bundle"](m, t, int(bool(rotate_points)), C.byref(view_in), C.byref(view_out),
                                                   n, self.stream_ptr()), "art_transform_bundle")

    def detector(self, ddesc, view, n, p3=None, XY=None, opl=None):
        p = [t.data_ptr() for t in p3] if p3 is not None else [None, None, None]
        xy = [t.data_ptr() for t in XY] if XY is not None else [None, None]
        o = opl.data_ptr() if opl is not None else None
        self.check(self.fn["art_detector"](C.byref(ddesc), C.byref(view), n, p[0], p[1], p[2], xy[0], xy[1], o,
                                           self.stream_ptr()), "art_detector")

    def detector_readout(self, ddesc, view, w, n, centres=(0.0, 0.0, 0.0), p3=None, XY=None, opl=None, to_host=True):
        """Fused read-out + statistics (art_detector_readout); returns the 24 statistics."""
        out = self.empty(24)      # n == 0: the library writes the reduction identities (0, +inf, -inf)
        p = [t.data_ptr() for t in p3] if (p3 is not None and n > 0) else [None, None, None]
        xy = [t.data_ptr() for t in XY] if (XY is not None and n > 0) else [None, None]
        sp, scratch = self.stream_ptr(), self._red_scratch()
        self.check(self._timed(lambda: self.fn["art_detector_readout"](
            C.byref(ddesc), C.byref(view), None if (w is None or n == 0) else w.data_ptr(), n, float(centres[0]),
            float(centres[1]), float(centres[2]), p[0], p[1], p[2], xy[0], xy[1],
            None if (opl is None or n == 0) else opl.data_ptr(), scratch.data_ptr(), out.data_ptr(), sp),
            "readout_events"), "art_detector_readout")
        return out.cpu().numpy() if to_host else out

    def detector_scan_moments(self, ddesc, view, w, n, co, span=0.0):
        """32 moment sums for a detector scan along its normal + [32] the number of rays whose path is not linear
        over shifts in [0, span] (art_detector_scan_moments); host array."""
        if n == 0:
            return np.zeros(33)
        out = self.empty(33)
        self.check(self.fn["art_detector_scan_moments"](C.byref(ddesc), C.byref(view),
                                                        None if w is None else w.data_ptr(), n, float(co), float(span),
                                                        self._red_scratch().data_ptr(), out.data_ptr(),
                                                        self.stream_ptr()), "art_detector_scan_moments")
        return out.cpu().numpy()

    def _red_scratch(self):
        return self.scratch("red", self.fn["art_reduce_scratch_doubles"](), torch.float64)

    def detector_stats(self, alive, X, Y, opl, w, n, to_host=True):
        out = self.empty(16)      # n == 0: one workgroup with nothing to add leaves the reduction identities
        ptr = lambda t: None if (t is None or n == 0) else t.data_ptr()
        if n == 0:
            alive = self.zeros(1, torch.uint8)
        self.check(self.fn["art_detector_stats"](alive.data_ptr(), ptr(X), ptr(Y), ptr(opl), ptr(w), n,
                                                 self._red_scratch().data_ptr(), out.data_ptr(), self.stream_ptr()),
                   "art_detector_stats")
        return out.cpu().numpy() if to_host else out

    def detector_moments(self, alive, X, Y, opl, w, n, cx, cy, co):
        if n == 0:
            return np.zeros(8)
        out = self.empty(8)
        ptr = lambda t: None if t is None else t.data_ptr()
        self.check(self.fn["art_detector_moments"](alive.data_ptr(), ptr(X), ptr(Y), ptr(opl), ptr(w), n,
                                                   float(cx), float(cy), float(co), self._red_scratch().data_ptr(),
                                                   out.data_ptr(), self.stream_ptr()), "art_detector_moments")
        return out.cpu().numpy()

    def bundle_sums(self, view, w, n):
        if n == 0:
            return np.zeros(8)
        out = self.empty(8)
        self.check(self.fn["art_bundle_sums"](C.byref(view), None if w is None else w.data_ptr(), n,
                                              self._red_scratch().data_ptr(), out.data_ptr(), self.stream_ptr()),
                   "art_bundle_sums")
        return out.cpu().numpy()

    def gaussian_intensity(self, view, axis, fraction, n):
        w = self.empty(n)
        a = (C.c_double * 3)(*[float(v) for v in axis])
        self.check(self.fn["art_gaussian_intensity"](C.byref(view), a, float(fraction), n,
                                                     self._red_scratch().data_ptr(), w.data_ptr(), self.stream_ptr()),
                   "art_gaussian_intensity")
        return w

    def gaussian_intensity_central(self, view, fraction, n):
        """Gaussian weights about the bundle's own central ray, axis formed on the device (art_gaussian_intensity_central);
        nothing returns to the host."""
        w = self.empty(n)
        sums8 = self.empty(8)
        self.check(self.fn["art_gaussian_intensity_central"](C.byref(view), float(fraction), n, self._red_scratch().data_ptr(),
                                                             sums8.data_ptr(), w.data_ptr(), self.stream_ptr()),
                   "art_gaussian_intensity_central")
        return w

    def bundle_max_angle(self, view, axis, n):
        """(largest angle to `axis`, largest |point|) over the alive rays; host floats."""
        if n == 0:
            return 0.0, 0.0
        out = self.empty(2)
        a = (C.c_double * 3)(*[float(v) for v in axis])
        self.check(self.fn["art_bundle_max_angle"](C.byref(view), a, n, self._red_scratch().data_ptr(),
                                                   out.data_ptr(), self.stream_ptr()), "art_bundle_max_angle")
        o = out.cpu().numpy()
        return float(o[0]), float(o[1])

    def compact(self, alive, n):
        """Returns (idx tensor int64 [count], count)."""
        if n == 0:
            return torch.empty(0, dtype=torch.int64, device=self.device), 0
        ints = self.fn["art_compact_scratch_ints"](n)
        sc = self.scratch("compact", ints, torch.int32)
        idx = torch.empty(int(max(n, 1)), dtype=torch.int64, device=self.device)
        cnt = torch.zeros(1, dtype=torch.int64, device=self.device)
        self.check(self.fn["art_compact"](alive.data_ptr(), n, sc.data_ptr(), idx.data_ptr(), cnt.data_ptr(),
                                          self.stream_ptr()), "art_compact")
        c = int(cnt.cpu().item())
        return idx[:c], c

    def make_source(self, kind, size, rot, S, first, n, n_total, view, step=1):
        """Slots 0..n-1 = global rays first, first + step, ... of an n_total-ray source (art_make_source_strided)."""
        r = (C.c_double * 9)(*[float(v) for v in np.asarray(rot).reshape(9)])
        s = (C.c_double * 3)(*[float(v) for v in np.asarray(S).reshape(3)])
        self.check(self.fn["art_make_source_strided"](kind, float(size), r, s, first, int(step), n, n_total, C.byref(view),
                                                      self.stream_ptr()), "art_make_source_strided")

    def exchange_pack(self, stats, X, Y, opl, alive, slots, send):
        """stats[24] + (X, Y, opl, alive) of the sampled slots -> send[24 + 4k] (art_exchange_pack)."""
        k = int(slots.numel())
        self.check(self.fn["art_exchange_pack"](stats.data_ptr(), X.data_ptr(), Y.data_ptr(), opl.data_ptr(),
                                                alive.data_ptr(), slots.data_ptr(), k, send.data_ptr(),
                                                self.stream_ptr()), "art_exchange_pack")

    def exchange_fold(self, recv, world, stride, out):
        """Fold `world` gathered statistics vectors into the global 24 (art_exchange_fold)."""
        self.check(self.fn["art_exchange_fold"](recv.data_ptr(), int(world), int(stride), out.data_ptr(),
                                                self.stream_ptr()), "art_exchange_fold")

    def survivor_bytes(self, count, dense=False):
        return int(self.fn["art_survivor_bytes"](int(count), 1 if dense else 0))

    def pack_survivors(self, alive, X, Y, opl, number, first, step, send):
        """Records (X, Y, path, number:int32) of the alive slots, in slot order, behind a (count, flags) header in
        the uint8 tensor `send` (art_pack_survivors); nothing returns to the host."""
        n = int(alive.numel())
        sc = self.scratch("compact", self.fn["art_compact_scratch_ints"](n), torch.int32)
        ptr = lambda t: None if (t is None or n == 0) else t.data_ptr()
        self.check(self.fn["art_pack_survivors"](ptr(alive), n, ptr(X), ptr(Y), ptr(opl), ptr(number), int(first),
                                                 int(step), sc.data_ptr(), send.data_ptr(), int(send.numel()),
                                                 self.stream_ptr()), "art_pack_survivors")

    def survivor_finish(self, stats_dev, n, send, xhdr=None):
        """Header of a ZERO-COPY send buffer whose sections the read-out wrote directly (art_survivor_finish): (n, dense) if
        every slot is alive, else (count, unpacked); `xhdr` (26 doubles): the rank's block of the header exchange."""
        self.check(self.fn["art_survivor_finish"](stats_dev.data_ptr(), int(n), send.data_ptr(),
                                                  None if xhdr is None else xhdr.data_ptr(), self.stream_ptr()),
                   "art_survivor_finish")

    def survivor_xheader(self, send, stats_dev, xhdr):
        """The rank's block of the header exchange behind art_pack_survivors (art_survivor_xheader)."""
        self.check(self.fn["art_survivor_xheader"](send.data_ptr(), None if stats_dev is None else stats_dev.data_ptr(),
                                                   xhdr.data_ptr(), self.stream_ptr()), "art_survivor_xheader")

    def trace_guides(self, descs, rays, alive):
        """Advance guide ray j (row j of the DEVICE tensor rays[count, 8], in place) through descs[j]; alive[count] uint8
        (art_trace_guides, 8 rays per launch)."""
        count = len(descs)
        sp = self.stream_ptr()
        for k0 in range(0, count, _abi.ART_GUIDES_MAX):
            m = min(_abi.ART_GUIDES_MAX, count - k0)
            darr = (_abi.ArtElementDesc * m)(*descs[k0:k0 + m])
            self.check(self.fn["art_trace_guides"](darr, m, rays.data_ptr() + 64 * k0, alive.data_ptr() + k0, sp),
                       "art_trace_guides")

    def analyse_bundles(self, jobs, n):
        """art_analyse_bundles for a list of ArtAnalysisJob (host structs): uploads the job table, enqueues the four
        launches and returns the DEVICE tensor out[len(jobs), 64] -- nothing is read back here."""
        c = len(jobs)
        arr = (_abi.ArtAnalysisJob * c)(*jobs)
        nb = C.sizeof(arr)
        # pinned staging + device table from a per-size pool (pinning host memory costs milliseconds): the pinned image is
        # rewritten only after its previous upload has completed, the device table is read by launches enqueued before
        # the next upload on the same stream
        pkey = (nb, self.stream_key())
        pair = self._job_pool.get(pkey)
        if pair is None:
            pair = self._job_pool[pkey] = [torch.empty(nb, dtype=torch.uint8, pin_memory=True),
                                           torch.empty(nb, dtype=torch.uint8, device=self.device), None]
        host, dev, ev = pair
        if ev is not None:
            ev.synchronize()
        C.memmove(host.data_ptr(), C.addressof(arr), nb)
        dev.copy_(host, non_blocking=True)
        pair[2] = torch.cuda.Event()
        pair[2].record()
        out = torch.empty((c, _abi.ART_ANALYSIS_DOUBLES), dtype=torch.float64, device=self.device)
        # (jobs that bring their sums along need no per-tile partials, but the area is sized for the general case)
        scratch = self.scratch("analysis", self.fn["art_analysis_scratch_doubles"](c, int(n)), torch.float64)
        self.check(self.fn["art_analyse_bundles"](dev.data_ptr(), arr, c, int(n), scratch.data_ptr(), out.data_ptr(),
                                                  self.stream_ptr()), "art_analyse_bundles")
        return out

    def bundle_sums9(self, bundle):
        """DEVICE tensor [64] whose first nine doubles are the analysis sums of `bundle` (count, sum point, sum vector, sum
        intensity, sum path: art_analyse_bundles with ONE job of mode ART_JOB_SUMS -- the canonical fold order, the same
        bits as the analysis' own pass or the tail of a tracing launch); nothing is read back."""
        j = _abi.ArtAnalysisJob()
        j.b = bundle.view()
        j.w = None if bundle.intensity is None else bundle.intensity.data_ptr()
        j.mode = _abi.ART_JOB_SUMS
        return self.analyse_bundles([j], bundle.n_slots)[0]

    def make_extended_source(self, radius, divergence, n_points, per, rot, S, first, n, view):
        r = (C.c_double * 9)(*[float(v) for v in np.asarray(rot).reshape(9)])
        s = (C.c_double * 3)(*[float(v) for v in np.asarray(S).reshape(3)])
        self.check(self.fn["art_make_extended_source"](float(radius), float(divergence), int(n_points), int(per), r, s,
                                                       first, n, C.byref(view), self.stream_ptr()),
                   "art_make_extended_source")


def get_backend():
    """The process-wide backend; created on first use, raises loudly if the HIP path is unavailable."""
    global _BACKEND
    if _BACKEND is None:
        _BACKEND = HipBackend()
    return _BACKEND
