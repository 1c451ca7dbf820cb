"""RayBundle: the structure-of-arrays ray bundle that replaces the reference's `list[Ray]`
(ART/ModuleOpticalRay.py; survivor lists of ART/ModuleMirror.py:928-939).

Layout in HBM: one contiguous fp64 tensor `data[8, n]` (rows ox, oy, oz, dx, dy, dz, path, incidence; every
row is a unit-stride stream for the kernels) plus `alive[n]` (uint8).  n is the number of SOURCE rays and
stays fixed along a chain: slot i of every bundle is source ray i, so `number`, `intensity` and
`wavelength` are shared, not copied.  The list-of-survivors view the reference API exposes
(`len()`, indexing, iteration in source order) is produced lazily by a stable device compaction."""

import os

import numpy as np
import torch

from . import _abi
from . import _lib
from .ModuleOpticalRay import Ray

import itertools

ROW_OX, ROW_OY, ROW_OZ, ROW_DX, ROW_DY, ROW_DZ, ROW_PATH, ROW_INC = range(8)
_SERIAL = itertools.count(1)   # identity of a bundle for change detection (id() values can be recycled)


class RayBundle:
    def __init__(self, data, alive, number=None, intensity=None, wavelength=None, parent=None, backend=None):
        self.data = data              # torch float64 [8, n]
        self.alive = alive            # torch uint8 [n]
        self.number = number          # torch int64 [n] or None (= slot index)
        self.intensity = intensity    # torch float64 [n] or None
        self.wavelength = wavelength  # float or None (uniform, as every reference source produces)
        self._parent = parent         # bundle this one was traced from (for Ray.path tuples); see `parent`
        self._parent_resolver = None  # lazy history (ModuleProcessing.LazyHistory): called once, sets the real parent
        self._backend = backend or _lib.get_backend()
        self._index = None
        self._count = None
        self.version = 0              # bumped when the arrays are modified in place
        self._serial = next(_SERIAL)
        # host array [n, k] or None: the path segments rays already carried when a bundle WITHOUT parent was built
        # from Ray objects whose `path` tuples had k > 1 entries (the device keeps only their sum)
        self.path_head = None
        self._fused_readout = None    # (detector key, version, result) of a read-out computed in the tracing launch
        self._fused_sums = None       # (version, weights identity, result) of the analysis sums formed by the tracing launch
        self._content = None          # (key, version it was valid for): see content_key()

    # ------------------------------------------------------------------ parent link
    @property
    def parent(self):
        """The bundle this one was traced from.  A bundle handed out by a lazy history was traced WITHOUT its
        intermediate bundles; the first look at its parent (Ray.path tuples, path_segments) materialises them."""
        if self._parent_resolver is not None:
            resolve, self._parent_resolver = self._parent_resolver, None
            resolve()
        return self._parent

    @parent.setter
    def parent(self, value):
        self._parent = value
        self._parent_resolver = None

    def fused_sums(self):
        """DEVICE tensor [>= 9] of this bundle's analysis sums if the tracing launch formed them and neither the bundle nor
        its weights changed since (ArtChainReadout.sums -> ArtAnalysisJob.sums), else None."""
        fs = getattr(self, "_fused_sums", None)
        if fs is None or fs[0] != self.version:
            return None
        w = self.intensity
        if (None if w is None else (w.data_ptr(), w._version)) != fs[1]:
            return None
        return fs[2]["sums_dev"]

    def _share_parent(self, other):
        """`other` (a view / copy of this bundle) has this bundle's parent -- also when that is still to be materialised.
        (`other` is referred to weakly: a closure over it, stored on it, would be a reference cycle that keeps its device
        arrays alive until the cyclic collector runs.)"""
        if self._parent_resolver is not None:
            import weakref
            wo = weakref.ref(other)

            def resolve(s=self, wo=wo):
                o = wo()
                if o is not None:
                    o._parent = s.parent
            other._parent_resolver = resolve

    # ------------------------------------------------------------------ backend / persistence
    @property
    def backend(self):
        """The compute backend; a bundle restored from an archive is moved to the device on first use."""
        if self._backend is None:
            be = _lib.get_backend()
            data, alive = self._rows(self.alive.numel(), be.device)
            data.copy_(self.data)
            alive.copy_(self.alive)
            self.data, self.alive = data, alive
            self.number = None if self.number is None else self.number.to(be.device)
            self.intensity = None if self.intensity is None else self.intensity.to(be.device)
            self._backend = be
        return self._backend

    def __getstate__(self):
        """Archive form (mp.save_compressed, ART/ModuleProcessing.py:612-625): plain host arrays."""
        host = lambda t: None if t is None else t.detach().cpu().numpy()
        return {"data": host(self.data), "alive": host(self.alive), "number": host(self.number),
                "intensity": host(self.intensity), "wavelength": self.wavelength, "parent": self.parent,
                "version": self.version, "path_head": self.path_head}

    def __setstate__(self, st):
        t = lambda a: None if a is None else torch.from_numpy(np.ascontiguousarray(a))
        self.data, self.alive = t(st["data"]), t(st["alive"])
        self.number, self.intensity = t(st["number"]), t(st["intensity"])
        self._parent_resolver = None
        self.wavelength, self.parent, self.version = st["wavelength"], st["parent"], st["version"]
        self.path_head = st.get("path_head")
        self._backend = None
        self._index = None
        self._count = None
        self._fused_readout = None
        self._fused_sums = None
        self._content = None
        self._serial = next(_SERIAL)

    # ------------------------------------------------------------------ construction
    # Every row of the SoA block starts on a 512-byte boundary (row pitch = n rounded up to 64 doubles, alive rows to
    # 512 bytes): with an odd ray count (PlaneWaveDisk emits N - 1 rays) unpadded rows start 8 bytes off a cache
    # line, every 512-byte wave access then straddles 5 lines instead of 4 and stores become partial-line writes --
    # measured 0.33 ms instead of 0.24 ms per 1e7-ray element trace.
    @staticmethod
    def _pitch(n, unit=64):
        # ART_PITCH_EXTRA (diagnostic, multiples of 512 B): extra padding per row, to move the rows of a block against
        # each other in the HBM channel interleave (measured: no effect)
        return max(unit, -(-int(n) // unit) * unit) + unit * int(os.environ.get("ART_PITCH_EXTRA", "0"))

    @classmethod
    def _rows(cls, n, device, count=None):
        """([count,] 8, n) fp64 view with aligned rows + ([count,] n) uint8 view with aligned rows."""
        n = int(n)
        pd, pa = cls._pitch(n), cls._pitch(n, 512)
        if count is None:
            return (torch.empty((8, pd), dtype=torch.float64, device=device)[:, :n],
                    torch.empty(pa, dtype=torch.uint8, device=device)[:n])
        lead = tuple(count) if isinstance(count, (tuple, list)) else (count,)
        return (torch.empty(lead + (8, pd), dtype=torch.float64, device=device)[..., :n],
                torch.empty(lead + (pa,), dtype=torch.uint8, device=device)[..., :n])

    @classmethod
    def allocate(cls, n, like=None, backend=None):
        be = backend or (like.backend if like is not None else _lib.get_backend())
        data, alive = cls._rows(n, be.device)
        if like is not None:
            return cls(data, alive, like.number, like.intensity, like.wavelength, like, be)
        return cls(data, alive, backend=be)

    @classmethod
    def allocate_many(cls, n, count, like, backend=None):
        """`count` bundles carved out of two allocations ([count, 8, n] fp64 + [count, n] uint8): the per-element
        history of one chain."""
        be = backend or like.backend
        data, alive = cls._rows(n, be.device, count)
        return [cls(data[k], alive[k], like.number, like.intensity, like.wavelength, like, be) for k in range(count)]

    @classmethod
    def allocate_grid(cls, n, chains, count, likes, backend=None):
        """[chains][count] bundles out of two allocations: the histories of a list of chains traced in one launch."""
        be = backend or likes[0].backend
        data, alive = cls._rows(n, be.device, (chains, count))
        return [[cls(data[c, k], alive[c, k], likes[c].number, likes[c].intensity, likes[c].wavelength, likes[c], be)
                 for k in range(count)] for c in range(chains)]

    @classmethod
    def from_arrays(cls, point, vector, number=None, intensity=None, wavelength=None, path0=None, backend=None):
        """Bundle from host (n,3) arrays: the two arrays are uploaded as they are and transposed to SoA and
        normalised on the device (art_pack_rays)."""
        be = backend or _lib.get_backend()
        point = np.ascontiguousarray(point, dtype=np.float64).reshape(-1, 3)
        vector = np.ascontiguousarray(vector, dtype=np.float64).reshape(-1, 3)
        n = point.shape[0]
        b = cls.allocate(n, backend=be)
        if n > 0:
            p0 = None
            if path0 is not None:
                p0 = be.from_numpy(np.broadcast_to(np.asarray(path0, dtype=np.float64), (n,)).copy())
            be.pack_rays(be.from_numpy(point), be.from_numpy(vector), p0, n, b.view())
        b.number = None if number is None else be.from_numpy(np.asarray(number, dtype=np.int64))
        b.intensity = None if intensity is None else be.from_numpy(np.asarray(intensity, dtype=np.float64))
        b.wavelength = wavelength
        return b

    @classmethod
    def from_ray_list(cls, rays, backend=None):
        rays = list(rays)
        pts = np.array([r.point for r in rays], dtype=np.float64).reshape(-1, 3)
        vec = np.array([r.vector for r in rays], dtype=np.float64).reshape(-1, 3)
        nums = [r.number for r in rays]
        number = None if any(v is None for v in nums) else np.array(nums, dtype=np.int64)
        ints = [r.intensity for r in rays]
        intensity = None if any(v is None for v in ints) else np.array(ints, dtype=np.float64)
        path0 = np.array([float(np.sum(r.path)) for r in rays])
        wl = rays[0].wavelength if rays else None
        b = cls.from_arrays(pts, vec, number, intensity, wl, path0, backend)
        k = {len(r.path) for r in rays}
        if len(k) == 1 and k != {1}:
            b.path_head = np.array([r.path for r in rays], dtype=np.float64)
        return b

    # ------------------------------------------------------------------ raw access
    @property
    def n_slots(self):
        return int(self.alive.numel())

    def view(self):
        p = self.data.data_ptr()
        stride = self.data.stride(0) * 8      # row pitch in bytes (rows are padded, see _rows)
        v = _abi.ArtBundleView()
        v.ox, v.oy, v.oz = p, p + stride, p + 2 * stride
        v.dx, v.dy, v.dz = p + 3 * stride, p + 4 * stride, p + 5 * stride
        v.path, v.incidence = p + 6 * stride, p + 7 * stride
        v.alive = self.alive.data_ptr()
        return v

    def slots(self, lo, hi):
        """A bundle OBJECT over slots [lo, hi) of this one's arrays (no copy: the rows are unit-stride streams, a range of
        slots is a range of every row).  For tiled launches: the tiles of one trace write disjoint ranges of the same
        output bundles.  `lo` should be a multiple of 64 so that every row of the range starts on a 512-byte boundary."""
        lo, hi = int(lo), int(hi)
        cut = lambda t: None if t is None else t[lo:hi]
        out = RayBundle(self.data[:, lo:hi], self.alive[lo:hi], cut(self.number), cut(self.intensity), self.wavelength, None,
                        self._backend)
        if self.number is None and lo != 0:
            out.number = torch.arange(lo, hi, dtype=torch.int64, device=self.alive.device)     # slot i of the range is ray lo + i
        return out

    # ------------------------------------------------------------------ content identity
    def tag_content(self, key):
        """Declare that this bundle's contents (ray state, numbers, intensities, wavelength) are a pure function of the
        hashable `key` -- e.g. the arguments of the generator that filled it.  Two bundles with equal keys are
        bit-identical; the tag dies with the next touch()."""
        self._content = (key, self.version)

    def content_key(self):
        """Hashable that is equal for two bundles only if their contents are known to be bit-identical (same
        generator arguments, or a copy of such a bundle); unique per bundle and version otherwise.  Lets a list of
        chains that start from equal sources share the trace of their common prefix (RayTracingCalculationMany)."""
        if self._content is not None and self._content[1] == self.version:
            return self._content[0]
        return ("bundle", self._serial, self.version)

    def touch(self):
        self.version += 1
        self._index = None
        self._count = None

    def index(self):
        """int64 tensor of the slots of surviving rays, in source order."""
        if self._index is None:
            self._index, self._count = self.backend.compact(self.alive, self.n_slots)
        return self._index

    def __len__(self):
        if self._count is None:
            self.index()
        return self._count

    # ------------------------------------------------------------------ survivor arrays on the host
    def _gather(self, row):
        return self.data[row].index_select(0, self.index()).cpu().numpy()

    def points(self):
        idx = self.index()
        return self.data[0:3].index_select(1, idx).cpu().numpy().T.copy()

    def vectors(self):
        idx = self.index()
        return self.data[3:6].index_select(1, idx).cpu().numpy().T.copy()

    def paths_total(self):
        return self._gather(ROW_PATH)

    def incidences(self):
        return self._gather(ROW_INC)

    def numbers(self):
        idx = self.index()
        if self.number is None:
            return idx.cpu().numpy()
        return self.number.index_select(0, idx).cpu().numpy()

    def intensities(self):
        if self.intensity is None:
            return None
        return self.intensity.index_select(0, self.index()).cpu().numpy()

    def path_segments(self):
        """(m, k+1) per-segment lengths of the survivors: the reference's Ray.path tuples
        (ModuleMirror.py:904, ModuleMask.py:100), rebuilt from the cumulative paths along the parent links."""
        idx = self.index()
        cum = []
        b = self
        while b is not None:
            b.backend                 # a bundle restored from an archive moves back to the device on first use
            cum.append(b.data[ROW_PATH].index_select(0, idx).cpu().numpy())
            b = b.parent
        cum = cum[::-1]
        head = self._head_segments(idx.cpu().numpy())
        segs = (list(head.T) if head is not None else [cum[0]]) + [cum[j] - cum[j - 1] for j in range(1, len(cum))]
        return np.stack(segs, axis=1)

    def _head_segments(self, slots):
        """Rows `slots` of the root bundle's path_head (None if the root carries single-entry paths)."""
        b = self
        while b.parent is not None:
            b = b.parent
        return None if b.path_head is None else b.path_head[slots]

    # ------------------------------------------------------------------ list-of-Ray protocol
    def _ray_at(self, slots):
        slots = np.atleast_1d(np.asarray(slots, dtype=np.int64))
        st = torch.from_numpy(slots).to(self.backend.device)
        cols = self.data.index_select(1, st).cpu().numpy()
        nums = slots if self.number is None else self.number.index_select(0, st).cpu().numpy()
        ints = None if self.intensity is None else self.intensity.index_select(0, st).cpu().numpy()
        cum = []
        b = self
        while b is not None:
            b.backend                 # (restored parents may still be host-resident)
            cum.append(b.data[ROW_PATH].index_select(0, st).cpu().numpy())
            b = b.parent
        cum = cum[::-1]
        head = self._head_segments(slots)
        rays = []
        for j in range(len(slots)):
            first = (float(cum[0][j]),) if head is None else tuple(float(v) for v in head[j])
            path = first + tuple(float(cum[k][j] - cum[k - 1][j]) for k in range(1, len(cum)))
            inc = cols[ROW_INC, j]
            rays.append(Ray(cols[0:3, j].copy(), cols[3:6, j].copy(), path, int(nums[j]), self.wavelength,
                            None if np.isnan(inc) else np.float64(inc),
                            None if ints is None else np.float64(ints[j])))
        return rays

    def __getitem__(self, i):
        idx = self.index()
        if isinstance(i, slice):
            sel = idx[i].cpu().numpy()
            return self._ray_at(sel) if len(sel) else []
        m = len(self)
        if i < 0:
            i += m
        if not 0 <= i < m:
            raise IndexError("list index out of range")
        return self._ray_at([int(idx[i].item())])[0]

    def __iter__(self):
        idx = self.index().cpu().numpy()
        for lo in range(0, len(idx), 4096):
            for r in self._ray_at(idx[lo:lo + 4096]):
                yield r

    def subset(self, positions):
        """New bundle in which only the survivors at the given positions stay alive (replaces
        `np.random.choice(RayList, k)`, ARTmain.py:168-171)."""
        idx = self.index()
        pos = torch.as_tensor(np.asarray(positions, dtype=np.int64), device=self.backend.device)
        alive = torch.zeros_like(self.alive)
        alive[idx.index_select(0, pos)] = 1
        out = RayBundle(self.data, alive, self.number, self.intensity, self.wavelength, self._parent, self.backend)
        self._share_parent(out)
        out.path_head = self.path_head
        return out

    def transformed(self, M, T, rotate_points=True):
        """Affine map of the whole bundle on the device (art_transform_bundle): point' = M point + T (or
        point + T), vector' = normalize(M vector).  Scene-manipulation helper, not on the tracing path."""
        out = RayBundle.allocate(self.n_slots, like=self, backend=self.backend)
        out._parent = self._parent
        self._share_parent(out)
        out.path_head = self.path_head
        self.backend.transform_bundle(M, T, rotate_points, self.view(), out.view(), self.n_slots)
        return out

    def copy(self):
        data, alive = self._rows(self.n_slots, self.data.device)
        data.copy_(self.data)
        alive.copy_(self.alive)
        out = RayBundle(data, alive, self.number, self.intensity, self.wavelength, self._parent, self.backend)
        self._share_parent(out)
        out.path_head = self.path_head
        out.tag_content(self.content_key())        # a copy is bit-identical to its original
        return out

    def alias(self):
        """A new bundle OBJECT over the same device arrays (attributes such as `intensity` can be replaced per object;
        the arrays themselves are shared and must be treated as immutable -- `copy()` gives private storage).  What the
        chains of an OEPlacement loop list hold of their common source."""
        out = RayBundle(self.data, self.alive, self.number, self.intensity, self.wavelength, self._parent, self.backend)
        self._share_parent(out)
        out.path_head = self.path_head
        out.tag_content(self.content_key())
        hit = getattr(self, "_sum_w", None)
        if hit is not None and hit[0] == self.version:
            out._sum_w = (out.version, hit[1])
        fs = self._fused_sums
        if fs is not None and fs[0] == self.version:       # the same arrays, the same weights: the same sums
            out._fused_sums = (out.version, fs[1], fs[2])
        return out

    def __deepcopy__(self, memo):
        return self.copy()

    def __hash__(self):
        return hash((self._serial, self.version))
