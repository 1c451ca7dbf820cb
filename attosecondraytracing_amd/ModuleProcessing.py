"""Ray-tracing driver and (pre-)alignment helpers, API of ART/ModuleProcessing.py.

`RayTracingCalculation` is the drop-in boundary of this package: same name, arguments and return structure
as ART/ModuleProcessing.py:250-313, but the per-ray Python loops are replaced by launches of the gfx950
kernels in libart_hip.so on device-resident SoA bundles (bundle.RayBundle).  There is no CPU path."""
import contextlib
import copy
import lzma
import os
import pickle
from datetime import datetime
from time import perf_counter

import numpy as np
import torch

from . import _abi
from . import _lib
from . import ModuleGeometry as mgeo
from . import ModuleMask as mmask
from . import ModuleOpticalElement as moe
from . import ModuleOpticalRay as mray
from . import ModuleSupport as msupp
from .bundle import RayBundle

# "chain": one fused launch for the whole chain (ray stays in registers, every bundle still written);
# "element": one launch per optical element.  Results are identical.
DEFAULT_TRACE_MODE = os.environ.get("ART_TRACE_MODE", "chain")


# ------------------------------------------------------------------------------------------- descriptors
_DESC_CACHE = {}
_DESC_F64_OFFSET = _abi.ArtElementDesc.fwd.offset
assert (_abi.ArtElementDesc.mp.offset - _DESC_F64_OFFSET) == 30 * 8 and _abi.ArtElementDesc.mp.size == 32


def element_descriptor(oe, IgnoreDefects=True, backend=None):
    """ArtElementDesc (include/art_hip.h) of one OpticalElement; returns (desc, keepalive) where keepalive
    holds device tensors the descriptor points to.  Cached per element object until its pose/parameters change."""
    key = (id(oe), bool(IgnoreDefects))
    h = hash(oe)
    hit = _DESC_CACHE.get(key)
    if hit is not None and hit[0] == h and hit[3] is oe:
        return hit[1], hit[2]
    d, keep = _build_descriptor(oe, IgnoreDefects, backend)
    if len(_DESC_CACHE) > 4096:
        _DESC_CACHE.clear()
    _DESC_CACHE[key] = (h, d, keep, oe)
    return d, keep


def _build_descriptor(oe, IgnoreDefects, backend):
    optic = oe.type
    kind = getattr(optic, "_abi_kind", None)
    tname = getattr(optic, "type", None)
    if kind is None or not isinstance(tname, str) or not ("Mirror" in tname or tname == "Mask"):
        raise NameError("I don`t recognize the type of optical element " + str(tname) + ".")
    d = _abi.ArtElementDesc()
    d.kind = int(kind)
    d.support_kind = int(optic.support._abi_kind)
    fwd, bwd = mgeo.frame_maps(oe.normal, oe.majoraxis)
    # the 34 doubles fwd[9] bwd[9] pos[3] centre[3] sp[6] mp[4] lie back to back behind the four leading int32: written
    # through ONE NumPy view of the struct (element-wise ctypes assignment from Python lists cost 20 us per descriptor,
    # and a loop list builds 60 of them)
    buf = np.frombuffer(d, dtype=np.float64, count=34, offset=_DESC_F64_OFFSET)
    buf[0:9] = fwd.reshape(9)
    buf[9:18] = bwd.reshape(9)
    buf[18:21] = oe.position                            # OEPlacement builds integer-typed positions
    buf[21:24] = optic.get_centre()
    sp = optic.support._abi_params()
    buf[24:24 + len(sp)] = sp
    mp_ = optic._abi_params()
    buf[30:30 + len(mp_)] = mp_
    # e.g. MirrorEllipsoidal parameters without a surface point under the off-axis angle: see RayTracingCalculation
    d.nonfinite = bool(d.kind not in (_abi.ART_PLANE, _abi.ART_MASK)
                       and not (np.isfinite(buf[18:24]).all() and np.isfinite(buf[30:34]).all()))
    keep = None
    d.n_defects = 0
    d.n_grid = 0
    d.flags = 0
    d.zern = None
    d.grid = None
    if hasattr(optic, "DeformationList") and len(optic.DeformationList) > 0:
        be = backend or _lib.get_backend()
        keep = []
        zern, grids = optic._zernike_defects(), optic._grid_defects()
        if grids and not IgnoreDefects:
            # same outcome as the reference: DeformedMirror.get_normal -> Fourrier.get_normal raises
            grids[0].get_normal(None)
        if zern:
            table, n_tables, recurrence = optic._abi_defect_table()
            t = be.from_numpy(table)
            keep.append(t)
            d.zern = t.data_ptr()
            d.n_defects = n_tables
            if recurrence:
                d.flags |= _abi.ART_FLAG_ZERN_RECURRENCE
        if grids:
            arr = (_abi.ArtGridDefect * len(grids))()
            rect = np.asarray(optic.support._CircumRect(), dtype=float)
            for g, D in zip(arr, grids):
                # The kernels clamp a look-up outside the map to its edge; the reference's interpolator raises there
                # (ART/ModuleDefects.py:108-110, RegularGridInterpolator with bounds_error=True).  Hits lie inside the
                # mirror's support, so the two only differ when the support is larger than the map: refuse that here.
                half = np.array([max(abs(D._X[0]), abs(D._X[-1])), max(abs(D._Y[0]), abs(D._Y[-1]))])
                if (rect / 2 > half * (1 + 1e-12)).any():
                    raise ValueError(f"One of the requested xi is out of bounds: the mirror support ({rect[0]:g} x "
                                     f"{rect[1]:g} mm) is larger than the Fourrier map ({2 * half[0]:g} x {2 * half[1]:g} mm)")
                fields, dev = D._abi_grid(be)
                keep.append(dev)
                for k_, v_ in fields.items():
                    setattr(g, k_, v_)
            raw = np.frombuffer(bytes(arr), dtype=np.uint8).copy()
            t = be.from_numpy(raw)
            keep.append(t)
            d.grid = t.data_ptr()
            d.n_grid = len(grids)
        if not IgnoreDefects:
            d.flags |= _abi.ART_FLAG_PERTURBED_NORMAL
    return d, keep


def _as_bundle(rays, backend=None):
    if isinstance(rays, RayBundle):
        return rays
    return RayBundle.from_ray_list(rays, backend)


_CHAIN_MAX = 8          # elements per fused launch (kChainMax in csrc/art_scene.h)


# ------------------------------------------------------------------------------------------- the hot path
def _attach_readout(bundle, detector, path_centre, res):
    """Remember a read-out that was computed in the same launch as `bundle`: Detector.readout(bundle) returns it
    instead of launching the read-out kernel, as long as the detector pose and the bundle are unchanged.
    (`res` must not point back at the bundle: a reference cycle would keep every step's 2.6 GB of history alive until
    the cyclic collector runs, and the caching allocator would hand out fresh, untouched memory for every trace.)"""
    res.pop("bundle", None)
    bundle._fused_readout = (detector._readout_key(path_centre), bundle.version, res)


def _attach_sums(bundle, res):
    """Remember the SUMS the tracing launch formed for `bundle` (pass (1) of the analysis: ArtChainReadout.sums): the
    analysis of this bundle skips its own pass over it as long as the bundle and its weights are unchanged."""
    w = bundle.intensity
    bundle._fused_sums = (bundle.version, None if w is None else (w.data_ptr(), w._version), res)


def RayTracingCalculation(source_rays, optical_elements, IgnoreDefects=True, mode=None, history=True, detector=None,
                          path_centre=0.0, readout_lite=False, sums=False):
    """Propagate `source_rays` through `optical_elements` (ART/ModuleProcessing.py:250-313).

    Returns a list with one RayBundle per element: the rays *after* that element, in the lab frame.  Each
    bundle behaves like the reference's list of surviving Ray objects (len, indexing, iteration) and keeps
    the full SoA state on the device.  With history=False only the last bundle is materialised (the others
    are None): an extension for callers that only analyse the final bundle.

    detector: a placed `Detector` known BEFORE the trace (manual placement, a re-trace, a scan).  Its read-out of the
    last bundle (ART/ModuleDetector.py:191-279) is then computed in the same launch, while every ray is still in
    registers (art_trace_chain_readout): `detector.readout(outs[-1])` and the `get_*` methods find it ready instead of
    re-reading the bundle.  Same values as the separate read-out (statistics to rounding: another summation order).
    readout_lite=True: the fused read-out reduces only what `Detector.get_Delays` / `get_PointList2D[Centre]` /
    `get_OpticalPaths` consume (count, sum of paths, bounding box, path range: 8 of the 22 statistics, no weights).
    sums=True (no detector known yet): the launch forms pass (1) of the analysis of the last bundle instead -- count, sum
    point, sum vector, sum intensity, sum path, what `Detector.autoplace` and the transmission need
    (ART/ModuleDetector.py:109-137) -- so that the analysis that follows reads the bundle once, not twice."""
    if isinstance(history, str):
        if history != "lazy":
            raise ValueError("history must be True, False or 'lazy'")
        return LazyHistory(source_rays, optical_elements, IgnoreDefects, mode, detector, path_centre)
    src = _as_bundle(source_rays)
    be = src.backend
    n = src.n_slots
    m = len(optical_elements)
    if m == 0:
        return []
    descs, keep = [], []
    for oe in optical_elements:
        d, k = element_descriptor(oe, IgnoreDefects, be)
        descs.append(d)
        keep.append(k)
    mode = mode or DEFAULT_TRACE_MODE
    if any(d.flags & _abi.ART_FLAG_ZERN_RECURRENCE for d in descs):
        mode = "element"      # Zernike orders above 16 run the recurrences per ray: a kernel of its own, one element per launch
    bad = next((k for k, d in enumerate(descs) if d.nonfinite), None)
    if bad is not None:
        # The reference meets a mirror with NaN/inf parameters in np.roots (ART/ModuleGeometry.py:84, :99), which
        # raises LinAlgError for the first ray that reaches it -- and stays silent when no ray gets that far.
        head = RayTracingCalculation(src, optical_elements[:bad], IgnoreDefects, mode, True) if bad else []
        if len(head[-1] if bad else src) > 0:
            raise np.linalg.LinAlgError("Array must not contain infs or NaNs")
        tail = RayBundle.allocate_many(n, m - bad, head[-1] if bad else src, be)
        prev = head[-1] if bad else src
        for b in tail:
            b.alive.zero_()
            b.parent = prev
            prev = b
        return (head + tail) if history else [None] * (m - 1) + [tail[-1]]
    if history:
        outs = RayBundle.allocate_many(n, m, src, be)
    else:
        outs = [None] * (m - 1) + [RayBundle.allocate(n, like=src, backend=be)]
    prev = src
    for b in outs:
        if b is not None:
            b.parent = prev
            b._keepalive = keep
            prev = b
    if mode == "chain":
        views = [b.view() if b is not None else _abi.ArtBundleView() for b in outs]
        if not history and m > _CHAIN_MAX:
            # one fused launch covers at most 8 elements; the bundle handed from one launch to the next needs
            # storage even when the caller wants no history: two scratch bundles, used alternately
            scratch = [RayBundle.allocate(n, like=src, backend=be) for _ in range(min(2, (m - 1) // _CHAIN_MAX))]
            for j, k in enumerate(range(_CHAIN_MAX - 1, m - 1, _CHAIN_MAX)):
                views[k] = scratch[j % len(scratch)].view()
            outs[-1]._keepalive = (keep, scratch)
        ro = None
        if detector is not None and 0 < n <= be.MAX_FUSED_READOUT_RAYS:
            detector._iscomplete()
            ro = be.new_chain_readout(detector._desc(), src.intensity, n, (0.0, 0.0, path_centre), lite=readout_lite)
        elif sums and detector is None and 0 < n <= be.MAX_FUSED_READOUT_RAYS and hasattr(be, "new_chain_sums"):
            ro = be.new_chain_sums(src.intensity, n)
        be.trace_chain(descs, src.view(), views, n, readout=ro)
        if ro is not None and ro.get("sums"):
            _attach_sums(outs[-1], ro)
        elif ro is not None:
            _attach_readout(outs[-1], detector, path_centre, ro)
    elif mode == "element":
        if not history and m > 1:
            # ping-pong through one scratch bundle, in place
            cur = RayBundle.allocate(n, like=src, backend=be)
            be.trace_element(descs[0], src.view(), cur.view(), n)
            for k in range(1, m - 1):
                be.trace_element(descs[k], cur.view(), cur.view(), n)
            be.trace_element(descs[m - 1], cur.view(), outs[-1].view(), n)
        else:
            vin = src.view()
            for k in range(m):
                vout = outs[k].view()
                be.trace_element(descs[k], vin, vout, n)
                vin = vout
    else:
        raise ValueError("mode must be 'chain' or 'element'")
    return outs


class _ParentResolver:
    """What a bundle handed out by a lazy history keeps in order to get its parents back: a WEAK reference to the history
    (a strong one would close a cycle bundle -> history -> bundle, and every traced bundle -- 650 MB per 1e7 rays -- would
    stay on the device until Python's cyclic collector happens to run) plus the recipe of the trace, so that the parents
    can still be re-traced when the caller kept the bundle and dropped the history."""
    __slots__ = ("hist", "bundle", "recipe")

    def __init__(self, hist, bundle):
        import weakref
        self.hist, self.bundle = weakref.ref(hist), weakref.ref(bundle)
        self.recipe = (hist._src, hist._els, hist._opts, hist._want, hist._scene_key)

    def __call__(self):
        h = self.hist()
        if h is None:
            b = self.bundle()
            if b is None:
                return
            src, els, (ign, mode, pc_), want, key = self.recipe
            h = LazyHistory(src, els, ign, mode, None, pc_, want, first=b)
            h._scene_key = key          # the scene the bundle was traced through, not today's
        h._materialise()


class LazyHistory:
    """`output_rays` with the per-element history materialised ON DEMAND (`history="lazy"`).

    `ARTmain.run_ART` analyses ONE bundle of the list `RayTracingCalculation` returns
    (`output_rays[ReflectionNumber]`, ART/ARTmain.py:254-255), yet a full trace writes one 65-byte record per ray and
    element -- for the write-bound kernels that IS the time.  A lazy history traces the chain once WITHOUT the
    intermediate bundles (the wanted bundle + the fused read-out of a known detector) and behaves like the reference's
    list of bundles: the first access to any other entry -- or to anything that needs the parents of the wanted bundle
    (`Ray.path` tuples) -- re-traces the chain with the full history, once, and checks that the bundle handed out
    earlier is bit-identical to the one of the full trace (same kernels, same inputs: by construction)."""

    def __init__(self, source, elements, IgnoreDefects=True, mode=None, detector=None, path_centre=0.0, want=-1, first=None):
        self._src, self._els = _as_bundle(source), list(elements)
        # what the bundle handed out below was traced through: poses / parameters of the elements (their hashes) and the
        # source's contents (its version).  A later re-trace for the rest of the history must see the same scene.
        self._scene_key = (_hash_list_of_objects(self._els), hash(self._src))
        self._opts = (bool(IgnoreDefects), mode, path_centre)
        m = len(self._els)
        self._bundles = [None] * m
        self._full = m <= 1
        self.retraces = 0
        if m == 0:
            return
        self._want = want % m
        if first is None:
            # (no detector known: the launch forms the sums the analysis of this bundle starts from)
            first = RayTracingCalculation(self._src, self._els[:self._want + 1], IgnoreDefects, mode, False,
                                          detector if self._want == m - 1 else None, path_centre,
                                          sums=detector is None or self._want != m - 1)[-1]
        self._bundles[self._want] = first
        if not self._full:
            first._parent_resolver = _ParentResolver(self, first)

    def _materialise(self):
        if self._full:
            return
        if (_hash_list_of_objects(self._els), hash(self._src)) != self._scene_key:
            # the reference's history is computed eagerly and stays valid when the chain is modified afterwards; a lazy one
            # cannot be completed from a scene that has changed: say so instead of mixing two scenes in one history
            raise RuntimeError("lazy history: an optical element or the source bundle was modified after this history was "
                               "handed out; its remaining bundles can no longer be traced.  Ask for the full history "
                               "(get_output_rays() / history=True) before modifying the chain.")
        self._full = True
        ign, mode, pc_ = self._opts
        full = RayTracingCalculation(self._src, self._els, ign, mode, True, None, pc_)
        self.retraces += 1
        for k, b in enumerate(full):
            have = self._bundles[k]
            if have is None:
                self._bundles[k] = b
            else:
                # the bundle handed out before stays THE bundle (callers hold it); the full trace must reproduce it
                live = b.alive.bool()
                if not (torch.equal(have.alive, b.alive)
                        and torch.equal(have.data[:, live].view(torch.int64), b.data[:, live].view(torch.int64))):
                    raise RuntimeError("lazy history: the re-trace differs from the bundle handed out earlier")
                have.parent = self._src if k == 0 else self._bundles[k - 1]
        for k in range(1, len(full)):
            if self._bundles[k] is full[k]:
                self._bundles[k].parent = self._bundles[k - 1]

    def __len__(self):
        return len(self._bundles)

    def __getitem__(self, i):
        if isinstance(i, slice):
            self._materialise()
            return self._bundles[i]
        m = len(self._bundles)
        if not -m <= i < m:
            raise IndexError("list index out of range")
        if self._bundles[i % m] is None:
            self._materialise()
        return self._bundles[i % m]

    def __iter__(self):
        self._materialise()
        return iter(self._bundles)

    def __getstate__(self):
        st = dict(self.__dict__)
        for b in st["_bundles"]:
            if b is not None and b._parent_resolver is not None:      # an archive holds plain bundles
                self._materialise()
                return dict(self.__dict__)
        return st


def RayTracingCalculationMany(source_rays_list, optical_elements_list, IgnoreDefects=True, history=True,
                              detectors=None, sums=False):
    with mgeo.frozen_hashes():
        return _RayTracingCalculationMany(source_rays_list, optical_elements_list, IgnoreDefects, history, detectors, sums)


def _RayTracingCalculationMany(source_rays_list, optical_elements_list, IgnoreDefects=True, history=True,
                               detectors=None, sums=False):
    """`RayTracingCalculation` for a LIST of chains in ONE launch (art_trace_scene): what `OEPlacement` returns when one
    of its arguments is a list -- 10-11 chains that differ only in poses (ART/ModuleProcessing.py:203-239), which the
    reference's `ARTmain.main` traces one after the other (ARTmain.py:304-342).  The element descriptors of all chains
    travel as one device-resident scene table; blockIdx.y selects the chain.  Returns one list of bundles per chain,
    identical to separate calls.  Chains that cannot share a launch (different ray or element counts) or whose
    histories together exceed 4 GB are traced one by one -- still on the device; chains with equal sources share the
    trace of their common prefix (below).  `detectors`: one placed Detector per chain whose read-out is fused behind the
    trace (see RayTracingCalculation); `sums=True` (without detectors): every chain's launch forms pass (1) of the
    analysis of its last bundle instead."""
    sources = [_as_bundle(s) for s in source_rays_list]
    c = len(sources)
    if c != len(optical_elements_list):
        raise ValueError("need one element list per source bundle")
    if c == 0:
        return []
    m, n, be = len(optical_elements_list[0]), sources[0].n_slots, sources[0].backend
    descs, keep = [], []
    for els in optical_elements_list:
        for oe in els:
            d, k = element_descriptor(oe, IgnoreDefects, be)
            descs.append(d)
            keep.append(k)
    uniform = (m > 0 and n > 0 and all(len(els) == m for els in optical_elements_list)
               and all(s.n_slots == n and s.backend is be for s in sources) and not any(d.nonfinite for d in descs)
               and not any(d.flags & _abi.ART_FLAG_ZERN_RECURRENCE for d in descs))
    if detectors is not None and len(detectors) != c:
        raise ValueError("need one detector per chain")
    # Common prefix: a loop list varies ONE entry of one list (OEPlacement), so its chains start from equal sources and
    # share every element before the varied one -- C2 / C3: the mask and the first toroid, two of three elements.  That
    # prefix is traced ONCE and its bundles are shared by all chains (the same RayBundle objects in every chain's
    # result); only the suffixes go into the many-chain launch, all reading the prefix's last bundle.  Equality is
    # exact: descriptors byte for byte, sources by content key (same generator arguments / copies, bundle.content_key).
    if uniform and c > 1 and detectors is None and len({s.content_key() for s in sources}) == 1 \
            and sources[0].content_key()[0] != "bundle":
        L = 0
        while L < m and all(bytes(descs[ci * m + L]) == bytes(descs[L]) for ci in range(1, c)):
            L += 1
        if L > 0:
            head = RayTracingCalculation(sources[0], optical_elements_list[0][:L], IgnoreDefects, None, history,
                                         sums=sums and L == m)
            if L == m:
                return [list(head) for _ in range(c)]
            tails = RayTracingCalculationMany([head[-1]] * c, [els[L:] for els in optical_elements_list], IgnoreDefects,
                                              history, sums=sums)
            return [list(head) + t for t in tails]
    # One launch pays off where single launches are latency-bound (<= ~1e6 rays per chain).  With 1e7-ray chains the
    # kernels fill the GPU either way and ONE allocation for all histories (20 GB for C3) is slower to obtain than ten
    # 2-GB ones (tools/e2e_time.py: 10 chains x 1e7 rays 7.3 ms in one launch, 4.6 ms chain by chain; 1e5-1e6 rays
    # 3.9-4.0 ms both ways, host-bound by descriptor building): above 4 GB of history the chains are launched one by one.
    too_big = history and c * m * n * 65 > 4e9
    if c == 1 or not uniform or too_big:
        return [RayTracingCalculation(s, els, IgnoreDefects, None, history, None if detectors is None else detectors[k],
                                      sums=sums and detectors is None)
                for k, (s, els) in enumerate(zip(sources, optical_elements_list))]
    if history:
        grid = RayBundle.allocate_grid(n, c, m, sources, be)
    else:
        grid = [[None] * (m - 1) + [RayBundle.allocate(n, like=s, backend=be)] for s in sources]
    views, scratch = [], []
    for ci, outs in enumerate(grid):
        prev = sources[ci]
        for k, b in enumerate(outs):
            if b is None and (k + 1) % _CHAIN_MAX == 0:      # hand-over bundle between two fused launches
                b = RayBundle.allocate(n, like=sources[ci], backend=be)
                scratch.append(b)
                views.append(b.view())
                continue
            if b is not None:
                b.parent = prev
                b._keepalive = (keep, scratch)
                prev = b
            views.append(b.view() if b is not None else _abi.ArtBundleView())
    ros = None
    if detectors is not None and n <= be.MAX_FUSED_READOUT_RAYS:
        areas = be.chain_readout_scratch(n, c)
        ros = []
        for d, s_, area in zip(detectors, sources, areas):
            d._iscomplete()
            ros.append(be.new_chain_readout(d._desc(), s_.intensity, n, scratch=area))
    elif sums and detectors is None and n <= be.MAX_FUSED_READOUT_RAYS and hasattr(be, "new_chain_sums"):
        ros = [be.new_chain_sums(s_.intensity, n, scratch=area) for s_, area in zip(sources, be.chain_readout_scratch(n, c))]
    host, dev = be.scene_alloc(c, m, transient=True)
    be.scene_pack(descs, [s.view() for s in sources], views, c, m, host, ros)
    be.scene_upload(host, dev)
    be.trace_scene(dev, host, n, segments=-(-m // 8))
    for ci, outs in enumerate(grid):
        outs[-1]._keepalive = (keep, scratch)
        if ros is not None and ros[ci].get("sums"):
            _attach_sums(outs[-1], ros[ci])
        elif ros is not None:
            _attach_readout(outs[-1], detectors[ci], 0.0, ros[ci])
    return grid


# ------------------------------------------------------------------------------------------- placement
def _placement_source(SourceProperties, FirstOptic):
    """The source bundle `_singleOEPlacement` builds (ART/ModuleProcessing.py:32-130, its first half)."""
    from . import ModuleSource as msource

    Divergence = SourceProperties["Divergence"]
    SourceSize = SourceProperties["SourceSize"]
    RayNumber = SourceProperties["NumberRays"]
    Wavelength = SourceProperties["Wavelength"]
    SourcePosition = np.array([0, 0, 0])
    SourceDirection = np.array([1, 0, 0])
    if Divergence == 0:
        if SourceSize == 0:
            S0 = FirstOptic.support
            radius = 0.5 * min(S0.dimX, S0.dimY) if hasattr(S0, "dimX") else S0.radius
        else:
            radius = SourceSize / 2
        SourceRayList = msource.PlaneWaveDisk(SourcePosition, SourceDirection, radius, RayNumber,
                                              Wavelength=Wavelength)
    elif SourceSize == 0:
        SourceRayList = msource.PointSource(SourcePosition, SourceDirection, Divergence, RayNumber,
                                            Wavelength=Wavelength)
    else:
        SourceRayList = msource.ExtendedSource(SourcePosition, SourceDirection, SourceSize, Divergence, RayNumber,
                                               Wavelength=Wavelength)
    Source = msource.ApplyGaussianIntensityToRayList(SourceRayList, 1 / np.e ** 2)
    # The analysis of whatever is traced from this source needs its sum of intensities (the transmission's denominator,
    # ART/ModuleAnalysisAndPlots.py:62-77): formed HERE, while the device has nothing else to do (the placement that follows
    # is host arithmetic), and carried by the bundle like the sums a tracing launch forms (bundle.fused_sums) -- so the
    # analysis of a loop list does not spend a pass over the source on its critical path.
    be = Source.backend
    if hasattr(be, "bundle_sums9") and Source.n_slots > 0:
        _attach_sums(Source, {"sums_dev": be.bundle_sums9(Source)})
    return Source


def _placeChains(SourceProperties, OpticsList, variants, Description):
    """Place and align the optics of SEVERAL chains along their central rays (ART/ModuleProcessing.py:32-130 for one
    chain; :203-239 calls it once per value of a loop list).  `variants`: one (DistanceList, IncidenceAngleList,
    IncidencePlaneAngleList) per chain, all over the same OpticsList.

    The chains advance in lockstep, optic by optic: every chain's alignment ray lives in one small device array and ONE
    launch (art_trace_guides: guide j through element j) + ONE read-back moves them all across the optic just placed --
    instead of one trace of the growing guide chain, with its own read-back, per chain and mirror.  The rays' states
    are what those traces produce (the same per-ray code, element after element).  The source bundle is a pure function
    of SourceProperties and the first optic's support, hence the same for every chain of the list: it is generated
    ONCE, and every chain holds an alias of it (a bundle object of its own over the same immutable device arrays:
    bundle.RayBundle.alias)."""
    from . import ModuleOpticalChain as moc

    c = len(variants)
    Source = _placement_source(SourceProperties, OpticsList[0])
    be = Source.backend
    # The alignment rays live on a SIDE stream: their read-backs (one per mirror) then wait for a 64-thread kernel, not for
    # the source generation enqueued above on the caller's stream (0.5 ms of device work per 1e7 rays that the placement
    # arithmetic below overlaps instead of waiting for).  Only for optics without defect tables: those are uploaded
    # while their descriptor is built, and a table must not change streams between its upload and its readers.
    plain = not any(hasattr(O, "DeformationList") for O in OpticsList)
    guide_ctx = be.side_stream_context("guides") if (plain and hasattr(be, "side_stream_context")) else contextlib.nullcontext()
    with guide_ctx, mgeo.frozen_hashes():
        chains = _placeChainsOn(be, Source, OpticsList, variants, Description, c)
        if plain and hasattr(be, "side_stream_context"):
            torch.cuda.current_stream().synchronize()       # (the side stream: idle by now -- every read-back drained it)
    return chains


def _placeChainsOn(be, Source, OpticsList, variants, Description, c):
    from . import ModuleOpticalChain as moc
    plane_angles = [[np.deg2rad(a % 360) for a in v[2]] for v in variants]
    inc_angles = [[np.deg2rad(a % 360) for a in v[1]] for v in variants]
    centre = [np.array([0, 0, 0]) for _ in range(c)]
    central = [np.array([1, 0, 0]) for _ in range(c)]
    rot_axis = [np.array([0, 1, 0]) for _ in range(c)]   # normal of the incidence plane, initially the x-z plane
    elements = [[] for _ in range(c)]
    # one alignment ray per chain along the bundle axis: origin, direction, path, incidence (NaN: none yet)
    guides = be.from_numpy(np.tile(np.array([0.0, 0.0, 0.0, 1.0, 0.0, 0.0, 0.0, np.nan]), (c, 1)))
    guide_alive = be.from_numpy(np.ones(c, dtype=np.uint8))

    def advance(through, ready):
        """Every chain's guide ray through its element `through[j]` (`ready[j]`: its descriptor where the placement has
        built it already); chains whose element the guide kernel refuses (Zernike tables in the recurrence layout) go
        through the element kernel, one ray at a time."""
        descs = [d if d is not None else element_descriptor(oe, True, be)[0] for oe, d in zip(through, ready)]
        if any(d.nonfinite for d in descs):
            # a mirror with NaN / inf parameters: the reference's np.roots raises for the first ray that reaches it
            # (see RayTracingCalculation), and the guide ray always does
            raise np.linalg.LinAlgError("Array must not contain infs or NaNs")
        slow = [j for j, d in enumerate(descs) if d.flags & _abi.ART_FLAG_ZERN_RECURRENCE]
        if not slow:
            be.trace_guides(descs, guides, guide_alive)
            return
        host, al = guides.cpu().numpy(), guide_alive.cpu().numpy()
        for j in range(c):
            if not al[j]:
                continue
            one = RayBundle.from_arrays(host[j, 0:3], host[j, 3:6], path0=host[j, 6], backend=be)
            out = RayTracingCalculation(one, [through[j]])[-1]
            if len(out) == 0:
                al[j] = 0
            else:
                host[j] = out.data[:, 0].cpu().numpy()
        guides.copy_(be.from_numpy(host))
        guide_alive.copy_(be.from_numpy(al))

    # Every chain gets its OWN element objects and its own copies of the optics (the reference's OpticalChain deep-copies
    # its element list, ART/ModuleOpticalChain.py:88-95; an optic that appears twice in OpticsList stays ONE object inside
    # a chain, as deepcopy's memo keeps it): built here directly, so that the chains adopt them without a second copy.
    # The chains of a loop list are identical up to the varied entry: chains whose state in front of optic k is the same
    # (bit for bit: central ray, incidence plane, position, distance, angles) share ONE evaluation of the placement
    # arithmetic and ONE descriptor; the others repeat the reference's sequence (ART/ModuleProcessing.py:60-98).
    optic_copies = [{} for _ in range(c)]
    open_mask = None
    for k, Optic in enumerate(OpticsList):
        if not ("Mirror" in Optic.type or Optic.type == "Mask"):
            raise NameError("I don`t recognize the type of optical element " + Optic.type + ".")
        convex = Optic.type in ("SphericalCX Mirror", "CylindricalCX Mirror")
        shareable = not hasattr(Optic, "DeformationList")       # (defect tables belong to one optic object)
        groups, through, through_descs = {}, [], []
        for j in range(c):
            if convex:
                inc_angles[j][k] = np.pi - inc_angles[j][k]  # convex: reflect off the "back side"
            dist_ = variants[j][0][k]
            mine = optic_copies[j].get(id(Optic))
            if mine is None:
                mine = optic_copies[j][id(Optic)] = copy.deepcopy(Optic)
            key = (central[j].tobytes(), central[j].dtype.char, rot_axis[j].tobytes(), rot_axis[j].dtype.char, centre[j].tobytes(),
                   centre[j].dtype.char, type(dist_).__name__, float(dist_), plane_angles[j][k], inc_angles[j][k])
            g = groups.get(key)
            if g is None:
                ctr = central[j] * dist_ + centre[j]
                if abs(plane_angles[j][k] - np.pi) < 1e-10:
                    ra = -rot_axis[j]
                else:
                    ra = mgeo.RotationAroundAxis(central[j], -plane_angles[j][k], rot_axis[j])
                # (mgeo._cross3: np.cross for 3-vectors, the same products and differences, a tenth of its call overhead)
                normal = mgeo.RotationAroundAxis(ra, -np.pi / 2 + inc_angles[j][k], mgeo._cross3(central[j], ra))
                major = mgeo._cross3(ra, normal)
                element = moe.OpticalElement(mine, ctr, normal, major)       # (validates and normalises)
                g = groups[key] = {"ctr": ctr, "ra": ra, "first": element, "desc": None, "guide": None}
                if Optic.type == "Mask":
                    # the guide ray must always pass: a fully open stand-in mask (only for the guide)
                    if open_mask is None:
                        open_mask = mmask.Mask(msupp.SupportRoundHole(Radius=100, RadiusHole=100, CenterHoleX=0, CenterHoleY=0))
                    g["guide"] = moe.OpticalElement(open_mask, ctr, normal, major)
                if shareable:
                    d, keep = _build_descriptor(element, True, be)
                    g["desc"] = (hash(element), d, keep)
            else:
                element = moe.OpticalElement._like(g["first"], mine)          # own object and arrays, the same values
            centre[j], rot_axis[j] = g["ctr"], g["ra"]
            elements[j].append(element)
            if g["desc"] is not None:
                h, d, keep = g["desc"]
                if len(_DESC_CACHE) > 4096:
                    _DESC_CACHE.clear()
                _DESC_CACHE[(id(element), True)] = (h, d, keep, element)       # the trace finds it: equal content, equal hash
            guide_el = g["guide"] if g["guide"] is not None else element
            through.append(guide_el)
            through_descs.append(g["desc"][1] if (g["desc"] is not None and g["guide"] is None) else None)
        if k == len(OpticsList) - 1 and Optic.type == "Mask":
            break                              # nothing is placed behind it
        advance(through, through_descs)
        if "Mirror" in Optic.type:
            host, al = guides.cpu().numpy(), guide_alive.cpu().numpy()      # the one read-back of this optic
            seen = {}
            for j in range(c):
                if not al[j]:
                    raise IndexError("list index out of range")     # the reference indexes an empty survivor list here
                v = host[j, 3:6]
                kk = v.tobytes()
                u = seen.get(kk)
                if u is None:
                    u = seen[kk] = v / mgeo._norm(v)            # (the Ray.vector setter, ModuleOpticalRay.py:85-90)
                central[j] = u
    return [moc.OpticalChain._adopt(Source, els, Description) for els in elements]


def _singleOEPlacement(SourceProperties, OpticsList, DistanceList, IncidenceAngleList, IncidencePlaneAngleList,
                       Description):
    """Place and align the optics of one chain along the central ray (ART/ModuleProcessing.py:32-130)."""
    return _placeChains(SourceProperties, OpticsList, [(DistanceList, IncidenceAngleList, IncidencePlaneAngleList)],
                        Description)[0]


def _which_indeces(lst):
    return [i for i, x in enumerate(lst) if isinstance(x, (list, np.ndarray))]


def OEPlacement(SourceProperties, OpticsList, DistanceList, IncidenceAngleList, IncidencePlaneAngleList=None,
                Description="", *_ignored):
    """Automatic placement of the optics in the lab frame (ART/ModuleProcessing.py:133-246).  One entry of one
    of the three lists may itself be a list/array: then a list of OpticalChains is returned.

    Extra positional arguments are accepted and ignored: two shipped example configs pass a 7th `render`
    argument (examples/CONFIG_2toroidals_f-x-f.py:54, CONFIG_2toroidals_twisted.py:52)."""
    if IncidencePlaneAngleList is None:
        IncidencePlaneAngleList = np.zeros(len(OpticsList)).tolist()
    lists = {"incidence": IncidenceAngleList, "incplane": IncidencePlaneAngleList, "distance": DistanceList}
    nested = {k: _which_indeces(v) for k, v in lists.items()}
    total = sum(len(v) for v in nested.values())
    if total > 1:
        raise ValueError("Only one element of one of the lists IncidenceAngleList, IncidencePlaneAngleList, or "
                         "DistanceList can be a list or array itself. Otherwise things get too tangled...")
    if total == 0:
        return _singleOEPlacement(SourceProperties, OpticsList, DistanceList, IncidenceAngleList,
                                  IncidencePlaneAngleList, Description)
    labels = {"incidence": " incidence angle (deg)", "distance": " distance (mm)",
              "incplane": " incidence-plane angle rotation (deg)"}
    which = next(k for k in ("incidence", "distance", "incplane") if nested[k])
    i = nested[which][0]
    name = OpticsList[i].type + "_idx_" + str(i) + labels[which]
    loop_list = lists[which]
    values = copy.deepcopy(loop_list[i])
    variants = []
    for x in values:
        loop_list[i] = x
        variants.append((list(DistanceList), list(IncidenceAngleList), list(IncidencePlaneAngleList)))
    chains = _placeChains(SourceProperties, OpticsList, variants, Description)
    for ch, x in zip(chains, values):
        ch.loop_variable_name = name
        ch.loop_variable_value = x
    return chains


# ------------------------------------------------------------------------------------------- statistics
def FindCentralRay(RayList):
    """Mean point and mean direction of a bundle as a Ray (ART/ModuleProcessing.py:464-482)."""
    if isinstance(RayList, RayBundle):
        s = RayList.backend.bundle_sums(RayList.view(), None, RayList.n_slots)
        cnt = s[0]
        return mray.Ray(s[1:4] / cnt, s[4:7] / cnt)
    return mray.Ray(np.mean([x.point for x in RayList], axis=0), np.mean([x.vector for x in RayList], axis=0))


def StandardDeviation(List) -> float:
    """RMS spread of numbers or of points about their mean (ART/ModuleProcessing.py:485-507)."""
    A = np.asarray(List, dtype=float)
    if A.ndim == 1:
        return float(np.std(A))
    if A.ndim == 2 and A.shape[1] > 1:
        return float(np.sqrt(np.var(A, axis=0).sum()))
    raise ValueError("StandardDeviation expects a list of floats or numpy-arrays as input, but got something else.")


def WeightedStandardDeviation(List, Weights) -> float:
    """ART/ModuleProcessing.py:510-532."""
    A = np.asarray(List, dtype=float)
    average = np.average(A, axis=0, weights=Weights)
    variance = np.average((A - average) ** 2, axis=0, weights=Weights)
    return float(np.sqrt(np.sum(variance)))


def ReturnNumericalAperture(RayList, RefractiveIndex: float = 1) -> float:
    """n sin(theta_max) about the central ray (ART/ModuleProcessing.py:536-566); the max-angle reduction runs on
    the device for bundles."""
    central = FindCentralRay(RayList).vector
    if isinstance(RayList, RayBundle):
        amax, _ = RayList.backend.bundle_max_angle(RayList.view(), central, RayList.n_slots)
    else:
        amax = max(mgeo.AngleBetweenTwoVectors(central, r.vector) for r in RayList)
    return float(np.sin(amax) * RefractiveIndex)


def ReturnAiryRadius(Wavelength: float, NumericalAperture: float) -> float:
    """ART/ModuleProcessing.py:570-593."""
    if NumericalAperture > 1e-3 and Wavelength is not None:
        return 1.22 * 0.5 * Wavelength / NumericalAperture
    return 0


# ------------------------------------------------------------------------------------------- autofocus
def _spot_duration_at(M, shifts, weighted):
    """analysis.spot_duration_from_moments for MANY bundles and an array of shifts each: M [c, 33] moment rows, shifts [c, n]
    -> (spot sizes [c, n], durations [c, n]).  The same +, -, *, /, sqrt per element in the same order as the scalar
    function (exactly rounded operations: the same bits whether a bundle is evaluated alone or in a list)."""
    from .analysis import LightSpeed
    M = M[:, 16:32] if weighted else M[:, :16]
    m0 = M[:, 0].reshape(-1, 1, 1)
    c = M[:, 1:16].reshape(-1, 3, 5)                               # per bundle: rows X, Y, O; columns q, sq, qq, qs, ss
    q, sq, qq, qs, ss = (c[:, :, k:k + 1] for k in range(5))
    s = shifts[:, None, :]
    mean = (q + s * sq) / m0
    var = np.maximum((qq + 2 * s * qs + s * s * ss) / m0 - mean * mean, 0.0)
    return np.sqrt(var[:, 0] + var[:, 1]), np.sqrt(var[:, 2]) / LightSpeed * 1e15


def _optimise_many(items, OptFor, Amplitude, Precision, IntensityWeighted, verbose=False, announce=True):
    """The search of FindOptimalDistance (ART/ModuleProcessing.py:317-460) for MANY (Detector, RayList, analysis) triples at
    once: every scan level evaluates the positions of all bundles in one broadcast.  The read-out of every ray is linear
    in a shift of the detector along its normal, so spot size and duration at all positions follow from the ONE set of
    moment sums of each device analysis (analysis.BundleAnalysis, taken at its detector) instead of one pass over the
    bundle per position: arithmetic on 64 doubles per bundle.  Returns [(moved detector, spot size, duration)].

    Scan levels as in the reference: positions centre - A_k + i Step_k, i < int(2 A_k / Step_k) (19 or 20, by rounding:
    per bundle), A_k = Amplitude 0.1^k, k <= Precision; the best position of a level is the centre of the next.

    One set of moments serves every level.  The variances it yields, (qq + 2 s qs + s^2 ss) / m0 - mean^2, are quadratics
    in s evaluated near their minimum: their terms are of the size N X0^2 of the START pose and cancel down to N sigma^2 at
    the focus, so a variance carries an absolute error of ~1e-16 X0^2 -- against differences between neighbouring positions
    of the finest level of ~ss (1e-4 A)^2 ~ 1e-8 X0^2 (X0 ~ slope x amplitude): eight orders of room.  Shifting the
    moments to a level's centre first (q' = q + s0 sq, ...) would bake the same rounding into q', qq'; it is the quadratic
    itself that cancels.  tests/test_host_shell.py::test_autofocus_of_a_tight_focus checks the chosen positions of a
    sub-micrometre focus against a position-by-position evaluation.

    announce=False: the "no minimum in the searched range" remark is not printed here; every result gets a fourth entry
    (remark due?, lower end, upper end of the searched range) and the caller of a LIST prints under the chain it belongs to."""
    if OptFor not in ["intensity", "size", "duration"]:
        raise NameError("I don`t recognize what you want to optimize the detector distance for. OptFor must be "
                        "either 'intensity', 'size' or 'duration'.")
    c = len(items)
    first, amp = np.empty(c), np.empty(c)
    for j, (det, _, ana) in enumerate(items):
        first[j] = det.get_distance()
        if Amplitude is None:
            SizeSpot = 2 * ana.spot_duration(0.0, False)[0]
            NumericalAperture = float(np.sin(ana.max_angle) * 1)
            amp[j] = min(4 * np.ceil(SizeSpot / np.tan(np.arcsin(NumericalAperture))), first[j])
        else:
            amp[j] = Amplitude
    step = amp / 10
    if verbose:
        for j in range(c):
            print(f"Searching optimal detector position for *{OptFor}* within [{first[j]-amp[j]:.3f}, "
                  f"{first[j]+amp[j]:.3f}] mm...", end="", flush=True)
    if OptFor not in ("intensity", "duration", "spotsize"):
        # FindOptimalDistance lets "size" through, but the reference's scan only knows "spotsize": its fitness is
        # then never assigned (ART/ModuleProcessing.py:342-348)
        raise UnboundLocalError("local variable 'Fitness' referenced before assignment")
    M = np.stack([ana.moments for _, _, ana in items])
    shift = np.zeros(c)
    spot, dur = np.full(c, np.nan), np.full(c, np.nan)
    for k in range(Precision + 1):
        A, St = amp * 0.1 ** k, step * 0.1 ** k
        start = shift - A
        with np.errstate(invalid="ignore", divide="ignore"):
            ratio = 2 * A / St
        if not np.isfinite(ratio).all():
            raise ValueError("cannot convert float NaN to integer")        # (int(nan) in the scalar form: a zero amplitude)
        n = ratio.astype(np.int64)                                          # int(): truncation
        nmax = int(n.max())
        if nmax == 0:
            raise ValueError("attempt to get argmin of an empty sequence")  # (np.argmin([]) in the scalar form)
        shifts = start[:, None] + np.arange(nmax)[None, :] * St[:, None]
        sizes, durations = _spot_duration_at(M, shifts, IntensityWeighted)
        # A scan that carries the detector through the last optic (Amplitude clipped to the detector distance: the scan
        # starts AT the optic) meets rays whose hit lies behind their origin; the reference's path |I - A| has a kink
        # there and is no longer linear in the shift.  Those scans are evaluated position by position (still on the
        # device, still all rays), exactly as the reference's loop does.
        if OptFor in ("intensity", "duration"):
            for j, (det, rays, ana) in enumerate(items):
                if n[j] > 0 and not ana.linear_over(shifts[j, 0], shifts[j, n[j] - 1]):
                    for i in range(int(n[j])):
                        here = det.copy_detector()
                        here.shiftByDistance(float(shifts[j, i]))
                        sizes[j, i], durations[j, i] = here._spot_duration_from_moments(here._scan_moments(rays), 0.0,
                                                                                        IntensityWeighted)
        fitness = sizes ** 2 * durations if OptFor == "intensity" else (durations if OptFor == "duration" else sizes)
        fitness = np.where(np.arange(nmax)[None, :] < n[:, None], fitness, np.inf)
        if (n == 0).any():
            raise ValueError("attempt to get argmin of an empty sequence")
        ind = np.argmin(fitness, axis=1)
        rows = np.arange(c)
        shift = shifts[rows, ind]
        if OptFor in ("intensity", "spotsize"):
            spot = sizes[rows, ind]
        if OptFor in ("intensity", "duration"):
            dur = durations[rows, ind]
    results = []
    for j, (det, _, _) in enumerate(items):
        moving = det.copy_detector()
        moving.shiftByDistance(float(shift[j]))
        outside = not first[j] - amp[j] + 10 ** -Precision < moving.get_distance() < first[j] + amp[j] - 10 ** -Precision
        if outside and announce:
            print("There`s no minimum-size/duration focus in the searched range.")
        if verbose:
            print("\r\033[K", end="", flush=True)
        res = (moving, np.nan if OptFor == "duration" else float(spot[j]), float(dur[j]))
        results.append(res if announce else res + ((outside, first[j] - amp[j], first[j] + amp[j]),))
    return results


def _optimise_from_analysis(Detector, RayList, ana, OptFor, Amplitude, Precision, IntensityWeighted, verbose):
    """One bundle through _optimise_many (the same code path as a list: the same bits)."""
    out = _optimise_many([(Detector, RayList, ana)], OptFor, Amplitude, Precision, IntensityWeighted, verbose)[0]
    print("\r\033[K", end="", flush=True)
    return out


def FindOptimalDistance(Detector, RayList, OptFor="intensity", Amplitude: float = None, Precision: int = 3,
                        IntensityWeighted=False, verbose=False):
    """Detector distance minimising spot size, duration or spot^2*duration (ART/ModuleProcessing.py:369-460).
    Note: like the reference, the accepted names are 'intensity', 'size', 'duration' although the scan itself
    understands 'spotsize' (reference quirk, ModuleProcessing.py:424 vs :328).
    ONE device analysis of the bundle on `Detector` (the one `Detector.autoplace` left attached, if it still applies)
    serves the whole search: analysis.py."""
    if OptFor not in ["intensity", "size", "duration"]:
        raise NameError("I don`t recognize what you want to optimize the detector distance for. OptFor must be "
                        "either 'intensity', 'size' or 'duration'.")
    return _optimise_from_analysis(Detector, RayList, Detector._analysis_of(RayList), OptFor, Amplitude, Precision,
                                   IntensityWeighted, verbose)


# ------------------------------------------------------------------------------------------- misc
def _hash_list_of_objects(lst):
    return sum(hash(x) for x in lst)


def save_compressed(obj, filename: str = None):
    """lzma + pickle archive of results (ART/ModuleProcessing.py:612-625).  RayBundles are stored as host arrays."""
    if not isinstance(filename, str):
        filename = "kept_data_" + datetime.now().strftime("%Y-%m-%d-%Hh%M")
    i = 0
    while os.path.exists(filename + f"_{i}.xz"):
        i += 1
    filename = filename + f"_{i}"
    with lzma.open(filename + ".xz", "wb") as f:
        pickle.dump(obj, f)
    print("Saved results to " + filename + ".xz.")
    print("->To reload from disk do: kept_data = mp.load_compressed('" + filename + "')")


def load_compressed(filename: str):
    """ART/ModuleProcessing.py:628-633."""
    with lzma.open(filename + ".xz", "rb") as f:
        return pickle.load(f)


_tstart_stack = []


def _tic():
    _tstart_stack.append(perf_counter())


def _toc(fmt="Elapsed: %s s"):
    print(fmt % (perf_counter() - _tstart_stack.pop()))
