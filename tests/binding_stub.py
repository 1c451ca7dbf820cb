"""The reference-side binding of INTEGRATION.md (variant B), runnable: replaces the body of the reference's
`ModuleProcessing.RayTracingCalculation` by calls into the C ABI.  Works on the reference's OWN classes (it only reads
their public attributes).  `lib` / `prefix` select the library: libart_hip.so ("art_") on a GPU box, or the CPU twin
("art_cpu_") in the build container -- same structs, same entry points."""
import ctypes as C

import numpy as np
import torch

from attosecondraytracing_amd import _abi
from attosecondraytracing_amd.ModuleGeometry import frame_maps

KIND = {"Plane Mirror": 0, "SphericalCC Mirror": 1, "SphericalCX Mirror": 1, "Parabolic Mirror": 2,
        "Toroidal Mirror": 3, "Ellipsoidal Mirror": 4, "CylindricalCC Mirror": 5, "CylindricalCX Mirror": 5, "Mask": 6}


def _support(S):
    n = type(S).__name__
    if n == "SupportRound":
        return 0, [S.radius]
    if n == "SupportRoundHole":
        return 1, [S.radius, S.radiushole, S.centerholeX, S.centerholeY]
    if n == "SupportRectangle":
        return 2, [S.dimX, S.dimY]
    if n == "SupportRectangleHole":
        return 3, [S.dimX, S.dimY, S.radiushole, S.centerholeX, S.centerholeY]
    return 4, [S.dimX, S.dimY, S.holeX, S.holeY, S.centerholeX, S.centerholeY]


def _mirror_params(o):
    k = KIND[o.type]
    return {0: [], 1: lambda: [o.radius], 2: lambda: [o.p], 3: lambda: [o.majorradius, o.minorradius],
            4: lambda: [o.a, o.b], 5: lambda: [o.radius], 6: []}[k]() if k not in (0, 6) else []


def _desc(oe):
    d = _abi.ArtElementDesc()
    o = oe.type
    if o.type not in KIND:
        raise NameError("I don`t recognize the type of optical element " + o.type + ".")
    d.kind = KIND[o.type]
    d.support_kind, sp = _support(o.support)
    fwd, bwd = frame_maps(oe.normal, oe.majoraxis)
    d.fwd[:] = [float(v) for v in fwd.ravel()]
    d.bwd[:] = [float(v) for v in bwd.ravel()]
    d.pos[:] = [float(v) for v in oe.position]
    d.centre[:] = [float(v) for v in o.get_centre()]
    sp = [float(v) for v in sp]
    d.sp[:] = sp + [0.0] * (6 - len(sp))
    mp = [float(v) for v in _mirror_params(o)]
    d.mp[:] = mp + [0.0] * (4 - len(mp))
    return d


def _view(t, alive):
    v = _abi.ArtBundleView()
    p, s = t.data_ptr(), t.shape[1] * 8
    v.ox, v.oy, v.oz, v.dx, v.dy, v.dz, v.path, v.incidence = (p + k * s for k in range(8))
    v.alive = alive.data_ptr()
    return v


def make_binding(lib, prefix, device, Ray):
    """Returns a drop-in `RayTracingCalculation(source_rays, optical_elements, IgnoreDefects=True)`."""
    def fn(name):
        return getattr(lib, prefix + name)

    def RayTracingCalculation(source_rays, optical_elements, IgnoreDefects=True):
        n, m = len(source_rays), len(optical_elements)
        pts = torch.from_numpy(np.array([r.point for r in source_rays], dtype=float)).to(device)
        vec = torch.from_numpy(np.array([r.vector for r in source_rays], dtype=float)).to(device)
        p0 = np.array([float(np.sum(r.path)) for r in source_rays])
        path0 = torch.from_numpy(p0).to(device)
        src = torch.empty((8, n), dtype=torch.float64, device=device)
        alive = torch.empty(n, dtype=torch.uint8, device=device)
        extra = [] if prefix == "art_cpu_" else [None]                      # the stream argument of the HIP library
        vsrc = _view(src, alive)
        assert fn("pack_rays")(C.c_void_p(pts.data_ptr()), C.c_void_p(vec.data_ptr()), C.c_void_p(path0.data_ptr()),
                               C.c_int64(n), C.byref(vsrc), *extra) == 0
        outs = [(torch.empty_like(src), torch.empty_like(alive)) for _ in range(m)]
        descs = (_abi.ArtElementDesc * m)(*[_desc(oe) for oe in optical_elements])
        views = (_abi.ArtBundleView * m)(*[_view(t, a) for t, a in outs])
        rc = fn("trace_chain")(descs, C.c_int32(m), C.byref(vsrc), views, C.c_int64(n), *extra)
        if rc:
            raise RuntimeError("art_trace_chain failed: %d" % rc)
        result, cum = [], [p0]
        for t, a in outs:
            T, A = t.cpu().numpy(), a.cpu().numpy().astype(bool)
            cum.append(T[6])
            result.append([Ray(T[0:3, i].copy(), T[3:6, i].copy(),
                               tuple([float(cum[0][i])] + [float(cum[k][i] - cum[k - 1][i]) for k in range(1, len(cum))]),
                               source_rays[i].number, source_rays[i].wavelength, np.float64(T[7, i]),
                               source_rays[i].intensity) for i in np.nonzero(A)[0]])
        return result

    return RayTracingCalculation
