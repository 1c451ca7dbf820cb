"""TEST-ONLY backend: runs the host-side package against the CPU twin of the kernels (oracle/_twin/libart_twin.so,
same per-ray device functions compiled by g++) so that the API shell, descriptor packing and the kernel math can
be exercised in a container without a GPU.  The product never imports this; GPU tests use the real HipBackend."""
import ctypes as C
import os
import subprocess

import numpy as np
import torch

from attosecondraytracing_amd import _abi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TWIN = os.path.join(ROOT, "oracle", "_twin", "libart_twin.so")


def build_twin():
    # ART_TWIN_LIB: use another build of the twin, e.g. one made with -fsanitize=address,undefined (CPU only)
    if os.environ.get("ART_TWIN_LIB"):
        return os.environ["ART_TWIN_LIB"]
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle")], stdout=subprocess.DEVNULL,
                          stderr=subprocess.DEVNULL)
    return TWIN


class TwinBackend:
    name = "twin"

    def __init__(self):
        self.lib = C.CDLL(build_twin())
        self.device = torch.device("cpu")
        self.lib.art_cpu_trace_element.restype = C.c_int
        self.lib.art_cpu_trace_element.argtypes = [C.POINTER(_abi.ArtElementDesc), C.POINTER(_abi.ArtBundleView),
                                                   C.POINTER(_abi.ArtBundleView), C.c_int64]
        self.lib.art_cpu_trace_chain.restype = C.c_int
        self.lib.art_cpu_trace_chain.argtypes = [C.POINTER(_abi.ArtElementDesc), C.c_int32,
                                                 C.POINTER(_abi.ArtBundleView), C.POINTER(_abi.ArtBundleView),
                                                 C.c_int64]
        self.lib.art_cpu_detector.restype = C.c_int
        self.lib.art_cpu_detector.argtypes = [C.POINTER(_abi.ArtDetectorDesc), C.POINTER(_abi.ArtBundleView),
                                              C.c_int64] + [C.c_void_p] * 6
        self.lib.art_cpu_detector_scan.restype = C.c_int
        self.lib.art_cpu_detector_scan.argtypes = [C.POINTER(_abi.ArtDetectorDesc), C.POINTER(_abi.ArtBundleView),
                                                   C.c_int64, C.c_double] + [C.c_void_p] * 7
        self.lib.art_cpu_make_source.restype = C.c_int
        self.lib.art_cpu_make_source.argtypes = [C.c_int32, C.c_double, _abi.c_double_p, _abi.c_double_p, C.c_int64,
                                                 C.c_int64, C.c_int64, C.c_int64, C.POINTER(_abi.ArtBundleView)]
        self.lib.art_cpu_make_extended_source.restype = C.c_int
        self.lib.art_cpu_make_extended_source.argtypes = [C.c_double, C.c_double, C.c_int64, C.c_int64,
                                                          _abi.c_double_p, _abi.c_double_p, C.c_int64, C.c_int64,
                                                          C.POINTER(_abi.ArtBundleView)]

    def synchronize(self):
        pass

    def empty(self, n, dtype=torch.float64):
        return torch.empty(int(n), dtype=dtype)

    def zeros(self, n, dtype=torch.float64):
        return torch.zeros(int(n), dtype=dtype)

    def from_numpy(self, a, dtype=None):
        t = torch.from_numpy(np.array(a, copy=True))
        return t if dtype is None else t.to(dtype)

    def trace_element(self, desc, vin, vout, n):
        assert self.lib.art_cpu_trace_element(C.byref(desc), C.byref(vin), C.byref(vout), n) == 0

    MAX_FUSED_READOUT_RAYS = 1 << 28

    def new_chain_readout(self, ddesc, w, n, centres=(0.0, 0.0, 0.0), store=True, scratch=None, lite=False, targets=None):
        X, Y, opl = (torch.empty(n, dtype=torch.float64) for _ in range(3))     # the twin always computes them
        if targets is not None:
            X, Y, opl = targets
            store = True
        out = torch.empty(24, dtype=torch.float64)
        ro = _abi.ArtChainReadout()
        ro.det = ddesc
        ro.w = None if w is None else w.data_ptr()
        ro.cx, ro.cy, ro.co = (float(v) for v in centres)
        ro.X, ro.Y, ro.opl = X.data_ptr(), Y.data_ptr(), opl.data_ptr()
        ro.scratch, ro.out24 = out.data_ptr(), out.data_ptr()
        ro.lite = 1 if lite else 0
        return {"struct": ro, "X": X if store else None, "Y": Y if store else None, "opl": opl if store else None,
                "P3": None, "stats_dev": out, "_keep": (w, X, Y, opl), "_w": w, "_centres": centres, "lite": bool(lite)}

    def new_chain_sums(self, w, n, scratch=None):
        """ArtChainReadout.sums: the twin's trace leaves the sums to _finish_readout (NumPy, like its other reductions)."""
        out = torch.zeros(24, dtype=torch.float64)
        ro = _abi.ArtChainReadout()
        ro.w = None if w is None else w.data_ptr()
        ro.scratch, ro.out24 = out.data_ptr(), out.data_ptr()
        ro.lite, ro.sums = 0, 1
        return {"struct": ro, "sums_dev": out, "_keep": (w,), "_w": w, "sums": True}

    def chain_readout_scratch(self, n, count):
        return [None] * count

    def _finish_sums(self, ro, last_view, n):
        def arr(ptr, ty=C.c_double):
            return np.ctypeslib.as_array(C.cast(ptr, C.POINTER(ty)), shape=(max(n, 1),))[:n]
        a = arr(last_view.alive, C.c_uint8).astype(bool) if n else np.zeros(0, bool)
        out = np.zeros(24)
        out[0] = a.sum()
        for k, f_ in enumerate(("ox", "oy", "oz", "dx", "dy", "dz")):
            out[1 + k] = arr(getattr(last_view, f_))[a].sum() if n else 0.0
        w = ro["_w"]
        out[7] = w.numpy()[a].sum() if w is not None else a.sum()
        out[8] = arr(last_view.path)[a].sum() if n else 0.0
        ro["sums_dev"].copy_(torch.from_numpy(out))

    def _finish_readout(self, ro, last_view, n):
        """Statistics of a fused read-out (the twin's C side only fills X, Y, opl): same reduction as detector_readout."""
        if ro.get("sums"):
            return self._finish_sums(ro, last_view, n)
        X, Y, O = ro["_keep"][1:4]
        alive = torch.from_numpy(np.ctypeslib.as_array(C.cast(last_view.alive, C.POINTER(C.c_uint8)), shape=(max(n, 1),))[:n].copy())
        w, c = ro["_w"], ro["_centres"]
        out = np.zeros(24)
        out[:16] = self.detector_stats(alive, X, Y, O, w, n)
        a = alive.numpy().astype(bool)
        ww = w.numpy()[a] if w is not None else np.ones(int(a.sum()))
        ex, ey, eo = X.numpy()[a] - c[0], Y.numpy()[a] - c[1], O.numpy()[a] - c[2]
        out[16:22] = [(ex ** 2).sum(), (ey ** 2).sum(), (eo ** 2).sum(), (ww * ex ** 2).sum(), (ww * ey ** 2).sum(),
                      (ww * eo ** 2).sum()]
        if ro.get("lite"):          # ArtChainReadout.lite: only count, sum of paths, bounding box, path range
            out[6:12] = 0.0
            out[16:22] = 0.0
        ro["stats_dev"].copy_(torch.from_numpy(out))

    def trace_chain(self, descs, vin, vouts, n, readout=None):
        self._trace_chain(descs, vin, vouts, n)
        if readout is not None and readout.get("sums"):
            self._finish_sums(readout, vouts[len(descs) - 1], n)
        elif readout is not None:
            f = self.lib.art_cpu_chain_readout_tail
            f.restype, f.argtypes = C.c_int, [C.POINTER(_abi.ArtChainReadout), C.POINTER(_abi.ArtBundleView), C.c_int64]
            assert f(C.byref(readout["struct"]), C.byref(vouts[len(descs) - 1]), n) == 0
            self._finish_readout(readout, vouts[len(descs) - 1], n)

    def _trace_chain(self, descs, vin, vouts, n):
        m = len(descs)
        # the argument rules of art_trace_chain (csrc/art_kernels.hip), so that host-shell mistakes show up without a GPU:
        # the last view is mandatory, and a chain longer than one fused launch (8 elements) needs a view where one
        # launch hands over to the next
        if n > 0:
            assert vouts[m - 1].alive, "the last output view is mandatory"
            for k0 in range(0, m, 8):
                k1 = min(k0 + 8, m)
                assert vouts[k1 - 1].alive, "chains longer than 8 need a view every 8th element"
        darr = (_abi.ArtElementDesc * m)(*descs)
        varr = (_abi.ArtBundleView * m)(*vouts)
        assert self.lib.art_cpu_trace_chain(darr, m, C.byref(vin), varr, n) == 0

    # scene table: the same packer (csrc/art_scene.h) compiled into the twin, "device" image = the host image
    def scene_alloc(self, n_chains, n_elems, transient=False):
        f = self.lib.art_cpu_scene_bytes
        f.restype, f.argtypes = C.c_int64, [C.c_int32, C.c_int32]
        img = torch.empty(int(f(n_chains, n_elems)), dtype=torch.uint8)
        return img, img

    def scene_pack(self, descs, views_in, views_out, n_chains, n_elems, host_image, readouts=None):
        f = self.lib.art_cpu_scene_pack
        f.restype = C.c_int
        f.argtypes = [C.POINTER(_abi.ArtElementDesc), C.c_int32, C.c_int32, C.POINTER(_abi.ArtBundleView),
                      C.POINTER(_abi.ArtBundleView), C.POINTER(_abi.ArtChainReadout), C.c_void_p]
        darr = (_abi.ArtElementDesc * (n_chains * n_elems))(*descs)
        iarr = (_abi.ArtBundleView * n_chains)(*views_in)
        oarr = (_abi.ArtBundleView * (n_chains * n_elems))(*views_out)
        rarr = None if readouts is None else (_abi.ArtChainReadout * n_chains)(*[r["struct"] for r in readouts])
        rc = f(darr, n_chains, n_elems, iarr, oarr, rarr, host_image.data_ptr())
        assert rc >= 0, rc
        # the statistics of fused read-outs are finished on the Python side after every trace_scene of this image
        last = [views_out[c * n_elems + n_elems - 1] for c in range(n_chains)]
        self._scene_ro = getattr(self, "_scene_ro", {})
        self._scene_ro[host_image.data_ptr()] = None if readouts is None else list(zip(readouts, last))
        return rc

    def scene_upload(self, host_image, dev_image):
        return None

    def trace_scene(self, dev_image, host_image, n, segments=1):
        f = self.lib.art_cpu_trace_scene
        f.restype, f.argtypes = C.c_int, [C.c_void_p, C.c_void_p, C.c_int64]
        assert f(dev_image.data_ptr(), host_image.data_ptr(), n) == 0
        for ro, last in (getattr(self, "_scene_ro", {}).get(dev_image.data_ptr()) or []):
            self._finish_readout(ro, last, n)

    def trace_guides(self, descs, rays, alive):
        count = len(descs)
        darr = (_abi.ArtElementDesc * count)(*descs)
        f = self.lib.art_cpu_trace_guides
        f.restype, f.argtypes = C.c_int, [C.POINTER(_abi.ArtElementDesc), C.c_int32, C.c_void_p, C.c_void_p]
        assert f(darr, count, rays.data_ptr(), alive.data_ptr()) == 0

    def analyse_bundles(self, jobs, n):
        """Layout of art_analyse_bundles (include/art_hip.h): sums and moments reduced with NumPy, the detector placed
        by the same art_device.h code as on the device (art_cpu_analysis_place)."""
        place = self.lib.art_cpu_analysis_place
        place.restype = C.c_int
        place.argtypes = [_abi.c_double_p, C.c_int32, C.c_double, _abi.c_double_p, _abi.c_double_p, _abi.c_double_p,
                          _abi.c_double_p]
        scan = self.lib.art_cpu_detector_scan_kink
        scan.restype, scan.argtypes = C.c_int, [C.POINTER(_abi.ArtDetectorDesc), C.POINTER(_abi.ArtBundleView), C.c_int64] + [C.c_void_p] * 7
        out = np.zeros((len(jobs), _abi.ART_ANALYSIS_DOUBLES))

        def arr(ptr, ty=C.c_double):
            return np.ctypeslib.as_array(C.cast(ptr, C.POINTER(ty)), shape=(max(n, 1),))[:n]
        for j, jb in enumerate(jobs):
            o = out[j]
            o[53], o[54] = -np.inf, np.inf
            o[[56, 58, 60]], o[[57, 59, 61]] = np.inf, -np.inf
            v = jb.b
            a = arr(v.alive, C.c_uint8).astype(bool) if n else np.zeros(0, bool)
            w = arr(jb.w)[a] if (jb.w and n) else np.ones(int(a.sum()))
            if jb.sums:       # pass (1) came with the job (ArtAnalysisJob.sums): used as given
                o[:9] = np.ctypeslib.as_array(C.cast(jb.sums, C.POINTER(C.c_double)), shape=(9,))
            else:
                o[0] = a.sum()
                for k, f_ in enumerate(("ox", "oy", "oz", "dx", "dy", "dz")):
                    o[1 + k] = arr(getattr(v, f_))[a].sum() if n else 0.0
                o[7] = w.sum()
                o[8] = arr(v.path)[a].sum() if n else 0.0
            if jb.mode == _abi.ART_JOB_SUMS:
                continue
            if o[0] == 0:
                o[10:20] = np.nan
                continue
            sums9 = (C.c_double * 9)(*o[:9])
            res = (C.c_double * 22)()
            assert place(sums9, jb.mode, jb.distance, jb.centre, jb.normal, jb.refpoint, res) == 0
            res = np.array(res)
            o[10:13], o[13:16], o[16:19], o[19] = res[0:3], res[3:6], res[15:18], res[21]
            axis, co = res[18:21], res[21]
            d = _abi.ArtDetectorDesc()
            d.centre[:], d.normal[:], d.rot[:] = list(res[0:3]), list(res[3:6]), list(res[6:15])
            arrs = [np.zeros(n) for _ in range(7)]
            assert scan(C.byref(d), C.byref(v), n, *[x.ctypes.data for x in arrs]) == 0
            X, Y, O, sx, sy, so, sk = (x[a] for x in arrs)
            o[56:62] = [X.min(), X.max(), Y.min(), Y.max(), O.min(), O.max()]
            o[53] = sk[sk <= 0].max() if (sk <= 0).any() else -np.inf
            o[54] = sk[sk > 0].min() if (sk > 0).any() else np.inf
            V = np.stack([arr(v.dx), arr(v.dy), arr(v.dz)], axis=1)[a]
            u, vn = np.linalg.norm(axis), np.linalg.norm(V, axis=1)[:, None]
            o[55] = (2 * np.arctan2(np.linalg.norm(axis[None, :] * vn - V * u, axis=1),
                                    np.linalg.norm(axis[None, :] * vn + V * u, axis=1))).max()
            O = O - co
            so = so - 1.0
            for base, wt in ((0, np.ones_like(w)), (16, w)):
                o[20 + base] = wt.sum()
                for k, (q0, sq) in enumerate(((X, sx), (Y, sy), (O, so))):
                    p_ = 20 + base + 1 + 5 * k
                    o[p_:p_ + 5] = [(wt * q0).sum(), (wt * sq).sum(), (wt * q0 * q0).sum(), (wt * q0 * sq).sum(),
                                    (wt * sq * sq).sum()]
        return torch.from_numpy(out)

    def pack_rays(self, points, vectors, path0, n, view):
        f = self.lib.art_cpu_pack_rays
        f.restype = C.c_int
        f.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.POINTER(_abi.ArtBundleView)]
        assert f(points.data_ptr(), vectors.data_ptr(), None if path0 is None else path0.data_ptr(), n, C.byref(view)) == 0

    def transform_bundle(self, M, T, rotate_points, vin, vout, n):
        m = (C.c_double * 9)(*[float(v) for v in np.asarray(M).reshape(9)])
        t = (C.c_double * 3)(*[float(v) for v in np.asarray(T).reshape(3)])
        f = self.lib.art_cpu_transform_bundle
        f.restype = C.c_int
        f.argtypes = [_abi.c_double_p, _abi.c_double_p, C.c_int32, C.POINTER(_abi.ArtBundleView),
                      C.POINTER(_abi.ArtBundleView), C.c_int64]
        assert f(m, t, int(bool(rotate_points)), C.byref(vin), C.byref(vout), n) == 0

    def detector(self, ddesc, view, n, p3=None, XY=None, opl=None):
        p = [t.data_ptr() for t in p3] if p3 is not None else [None, None, None]
        xy = [t.data_ptr() for t in XY] if XY is not None else [None, None]
        o = opl.data_ptr() if opl is not None else None
        assert self.lib.art_cpu_detector(C.byref(ddesc), C.byref(view), n, p[0], p[1], p[2], xy[0], xy[1], o) == 0

    def detector_readout(self, ddesc, view, w, n, centres=(0.0, 0.0, 0.0), p3=None, XY=None, opl=None, to_host=True):
        X = XY[0] if XY is not None else torch.empty(n, dtype=torch.float64)
        Y = XY[1] if XY is not None else torch.empty(n, dtype=torch.float64)
        O = opl if opl is not None else torch.empty(n, dtype=torch.float64)
        self.detector(ddesc, view, n, p3, (X, Y), O)
        alive = torch.from_numpy(np.ctypeslib.as_array(C.cast(view.alive, C.POINTER(C.c_uint8)), shape=(max(n, 1),))[:n].copy())
        out = np.zeros(24)
        out[:16] = self.detector_stats(alive, X, Y, O, w, n)
        a = alive.numpy().astype(bool)
        ww = w.numpy()[a] if w is not None else np.ones(int(a.sum()))
        ex, ey, eo = X.numpy()[a] - centres[0], Y.numpy()[a] - centres[1], O.numpy()[a] - centres[2]
        out[16:22] = [(ex ** 2).sum(), (ey ** 2).sum(), (eo ** 2).sum(), (ww * ex ** 2).sum(), (ww * ey ** 2).sum(),
                      (ww * eo ** 2).sum()]
        return out if to_host else torch.from_numpy(out)

    def detector_scan_moments(self, ddesc, view, w, n, co, span=0.0):
        arrs = [np.zeros(n) for _ in range(7)]
        assert self.lib.art_cpu_detector_scan(C.byref(ddesc), C.byref(view), n, float(span),
                                              *[a.ctypes.data for a in arrs]) == 0
        a = np.ctypeslib.as_array(C.cast(view.alive, C.POINTER(C.c_uint8)), shape=(max(n, 1),))[:n].astype(bool)
        X, Y, O, sx, sy, so, cr = (v[a] for v in arrs)
        O = O - co
        so = so - 1.0
        ww = w.numpy()[a] if w is not None else np.ones(int(a.sum()))
        out = np.zeros(33)
        out[32] = cr.sum()
        for base, wt in ((0, np.ones_like(ww)), (16, ww)):
            out[base] = wt.sum()
            for k, (q0, sq) in enumerate(((X, sx), (Y, sy), (O, so))):
                o = base + 1 + 5 * k
                out[o:o + 5] = [(wt * q0).sum(), (wt * sq).sum(), (wt * q0 * q0).sum(), (wt * q0 * sq).sum(),
                                (wt * sq * sq).sum()]
        return out

    @staticmethod
    def _np(t):
        return None if t is None else t.numpy()

    def detector_stats(self, alive, X, Y, opl, w, n, to_host=True):
        a = alive.numpy().astype(bool)
        X, Y, opl, w = (self._np(t) for t in (X, Y, opl, w))
        z = np.zeros(int(a.sum()))
        x = X[a] if X is not None else z
        y = Y[a] if Y is not None else z
        o = opl[a] if opl is not None else z
        ww = w[a] if w is not None else np.ones_like(z)
        out = np.zeros(16)
        out[[2, 4, 12]], out[[3, 5, 13]] = np.inf, -np.inf       # reduction identities when nothing is alive
        if len(z):
            out[:14] = [len(z), o.sum(), x.min(), x.max(), y.min(), y.max(), x.sum(), y.sum(), ww.sum(),
                        (ww * x).sum(), (ww * y).sum(), (ww * o).sum(), o.min(), o.max()]
        return out if to_host else torch.from_numpy(out)

    def detector_moments(self, alive, X, Y, opl, w, n, cx, cy, co):
        a = alive.numpy().astype(bool)
        X, Y, opl, w = (self._np(t) for t in (X, Y, opl, w))
        ww = w[a] if w is not None else np.ones(int(a.sum()))
        out = np.zeros(8)
        out[0] = ww.sum()
        out[1] = (ww * (X[a] - cx) ** 2).sum() if X is not None else 0
        out[2] = (ww * (Y[a] - cy) ** 2).sum() if Y is not None else 0
        out[3] = (ww * (opl[a] - co) ** 2).sum() if opl is not None else 0
        out[4] = a.sum()
        return out

    def bundle_sums(self, view, w, n):
        def arr(ptr, ty=C.c_double):
            return np.ctypeslib.as_array(C.cast(ptr, C.POINTER(ty)), shape=(n,))
        a = arr(view.alive, C.c_uint8).astype(bool)
        out = np.zeros(8)
        out[0] = a.sum()
        for j, f in enumerate(("ox", "oy", "oz", "dx", "dy", "dz")):
            out[1 + j] = arr(getattr(view, f))[a].sum()
        out[7] = w.numpy()[a].sum() if w is not None else 0.0
        return out

    def gaussian_intensity(self, view, axis, fraction, n):
        def arr(ptr, ty=C.c_double):
            return np.ctypeslib.as_array(C.cast(ptr, C.POINTER(ty)), shape=(n,))
        a = arr(view.alive, C.c_uint8).astype(bool)
        V = np.stack([arr(view.dx), arr(view.dy), arr(view.dz)], axis=1)
        P = np.stack([arr(view.ox), arr(view.oy), arr(view.oz)], axis=1)
        axis = np.asarray(axis, float)
        u, v = np.linalg.norm(axis), np.linalg.norm(V, axis=1)[:, None]
        ang = 2 * np.arctan2(np.linalg.norm(axis[None, :] * v - V * u, axis=1), np.linalg.norm(axis[None, :] * v + V * u, axis=1))
        div = ang[a].max() if a.any() else 0.0
        k = -0.5 * np.log(fraction)
        if div > 1e-12:
            w = np.exp(-2 * (np.tan(ang) / div) ** 2 * k)
        else:
            d = np.linalg.norm(P, axis=1)
            w = np.exp(-2 * (d / d[a].max()) ** 2 * k)
        return torch.from_numpy(w)

    def bundle_max_angle(self, view, axis, n):
        def arr(ptr, ty=C.c_double):
            return np.ctypeslib.as_array(C.cast(ptr, C.POINTER(ty)), shape=(n,))
        a = arr(view.alive, C.c_uint8).astype(bool)
        V = np.stack([arr(view.dx), arr(view.dy), arr(view.dz)], axis=1)[a]
        P = np.stack([arr(view.ox), arr(view.oy), arr(view.oz)], axis=1)[a]
        axis = np.asarray(axis, float)
        u, v = np.linalg.norm(axis), np.linalg.norm(V, axis=1)[:, None]
        ang = 2 * np.arctan2(np.linalg.norm(axis[None, :] * v - V * u, axis=1), np.linalg.norm(axis[None, :] * v + V * u, axis=1))
        return (float(ang.max()), float(np.linalg.norm(P, axis=1).max())) if len(V) else (0.0, 0.0)

    def compact(self, alive, n):
        idx = torch.nonzero(alive, as_tuple=False).reshape(-1)
        return idx, int(idx.numel())

    def make_source(self, kind, size, rot, S, first, n, n_total, view, step=1):
        r = (C.c_double * 9)(*[float(v) for v in np.asarray(rot).reshape(9)])
        s = (C.c_double * 3)(*[float(v) for v in np.asarray(S).reshape(3)])
        assert self.lib.art_cpu_make_source(kind, float(size), r, s, first, int(step), n, n_total, C.byref(view)) == 0

    def exchange_pack(self, stats, X, Y, opl, alive, slots, send):
        k = int(slots.numel())
        send[:24] = stats
        if k:
            send[24:24 + 4 * k] = torch.stack([X[slots], Y[slots], opl[slots], alive[slots].to(torch.float64)], dim=1).reshape(-1)

    def exchange_fold(self, recv, world, stride, out):
        allv = torch.stack([recv.reshape(-1)[r * stride:r * stride + 24] for r in range(world)])      # (recv may start inside a block)
        res = allv.sum(dim=0)
        for s in (2, 4, 12):
            res[s] = allv[:, s].min()
        for s in (3, 5, 13):
            res[s] = allv[:, s].max()
        out.copy_(res)

    @staticmethod
    def survivor_bytes(count, dense=False):
        return (16 + int(count) * (24 if dense else 28) + 15) // 16 * 16

    def pack_survivors(self, alive, X, Y, opl, number, first, step, send):
        """Layout of art_pack_survivors (include/art_hip.h), written with NumPy."""
        n = int(alive.numel())
        a = alive.numpy().astype(bool)
        c = int(a.sum())
        dense = number is None and c == n
        buf = send.numpy()
        buf[:16].view(np.int64)[:] = [c, 1 if dense else 0]
        for k, t in enumerate((X, Y, opl)):
            buf[16 + 8 * c * k:16 + 8 * c * (k + 1)].view(np.float64)[:] = t.numpy()[a]
        if not dense:
            num = number.numpy()[a] if number is not None else first + np.nonzero(a)[0] * step
            buf[16 + 24 * c:16 + 28 * c].view(np.int32)[:] = num.astype(np.int32)

    def bundle_sums9(self, bundle):
        j = _abi.ArtAnalysisJob()
        j.b = bundle.view()
        j.w = None if bundle.intensity is None else bundle.intensity.data_ptr()
        j.mode = _abi.ART_JOB_SUMS
        return self.analyse_bundles([j], bundle.n_slots)[0]

    def survivor_finish(self, stats_dev, n, send, xhdr=None):
        """art_survivor_finish: the header of a zero-copy send buffer (+ the rank's block of the header exchange)."""
        c = int(stats_dev[0].item())
        send.numpy()[:16].view(np.int64)[:] = [c, 1 if c == n else 2]
        if xhdr is not None:
            self.survivor_xheader(send, stats_dev, xhdr)

    def survivor_xheader(self, send, stats_dev, xhdr):
        x = xhdr.numpy()
        x[:2].view(np.int64)[:] = send.numpy()[:16].view(np.int64)
        x[2:26] = 0.0 if stats_dev is None else stats_dev.numpy()[:24]

    def make_extended_source(self, radius, divergence, n_points, per, rot, S, first, n, view):
        r = (C.c_double * 9)(*[float(v) for v in np.asarray(rot).reshape(9)])
        s = (C.c_double * 3)(*[float(v) for v in np.asarray(S).reshape(3)])
        assert self.lib.art_cpu_make_extended_source(float(radius), float(divergence), int(n_points), int(per), r, s,
                                                     first, n, C.byref(view)) == 0


def install():
    """Make the twin the process-wide backend (tests; also the target of bench.py's ART_BENCH_BACKEND_HOOK test hook)."""
    from attosecondraytracing_amd import _lib
    _lib._BACKEND = TwinBackend()
    return _lib._BACKEND


def install_failing_on_rank_1():
    """Test hook target: rank 1 dies during set-up (the launcher must then end the other ranks instead of letting them
    wait in a collective)."""
    import os
    if os.environ.get("RANK") == "1":
        raise RuntimeError("rank 1 fails on purpose (launcher test)")
    return install()


def install_hanging_on_rank_1():
    """Test hook target: rank 1 never gets anywhere (the launcher's wall-clock limit must end the job)."""
    import os
    import time
    if os.environ.get("RANK") == "1":
        time.sleep(3600)
    return install()


def install_sleeping_on_rank_1():
    """Test hook target: rank 1 joins the process group but is late for the first collective by far more than the group's
    timeout (rank 0 must give up with an error, not wait)."""
    import os
    import time
    if os.environ.get("RANK") == "1":
        time.sleep(60)
    return install()
