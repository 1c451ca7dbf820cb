"""ctypes mirror of include/art_hip.h (structs, enums, prototypes).  Keep in sync with ART_ABI_VERSION."""
import ctypes as C

ART_ABI_VERSION = 11

ART_OK = 0
ART_ERR_BAD_ARG = -1
ART_ERR_UNSUPPORTED = -2
ART_ERR_HIP = -3
ART_ERR_NO_DEVICE = -4

# enum ArtOpticKind
ART_PLANE, ART_SPHERE, ART_PARABOLA, ART_TORUS, ART_ELLIPSOID, ART_CYLINDER, ART_MASK = range(7)
# enum ArtSupportKind
ART_SUP_ROUND, ART_SUP_ROUNDHOLE, ART_SUP_RECT, ART_SUP_RECTHOLE, ART_SUP_RECTRECTHOLE = range(5)

ART_FLAG_PERTURBED_NORMAL = 1
ART_FLAG_ZERN_RECURRENCE = 2
ART_ZERN_RECURRENCE_MAX_ORDER = 64

ART_ZERN_MAX_ORDER = 16
ART_ZERN_DIM = ART_ZERN_MAX_ORDER + 1
ART_ZERN_STRIDE = 2 + 3 * ART_ZERN_DIM * ART_ZERN_DIM
ART_MAX_DEFECTS = 16

c_double_p = C.POINTER(C.c_double)
c_uint8_p = C.POINTER(C.c_uint8)


class ArtGridDefect(C.Structure):
    _fields_ = [("h", C.c_void_p), ("nx", C.c_int32), ("ny", C.c_int32),
                ("x0", C.c_double), ("y0", C.c_double), ("dx", C.c_double), ("dy", C.c_double)]


class ArtElementDesc(C.Structure):
    _fields_ = [
        ("kind", C.c_int32),
        ("support_kind", C.c_int32),
        ("n_defects", C.c_int32),
        ("flags", C.c_uint32),
        ("fwd", C.c_double * 9),
        ("bwd", C.c_double * 9),
        ("pos", C.c_double * 3),
        ("centre", C.c_double * 3),
        ("sp", C.c_double * 6),
        ("mp", C.c_double * 4),
        ("zern", C.c_void_p),
        ("grid", C.c_void_p),
        ("n_grid", C.c_int32),
        ("reserved", C.c_int32),
    ]


class ArtBundleView(C.Structure):
    _fields_ = [
        ("ox", C.c_void_p), ("oy", C.c_void_p), ("oz", C.c_void_p),
        ("dx", C.c_void_p), ("dy", C.c_void_p), ("dz", C.c_void_p),
        ("path", C.c_void_p),
        ("incidence", C.c_void_p),
        ("alive", C.c_void_p),
    ]


class ArtDetectorDesc(C.Structure):
    _fields_ = [
        ("centre", C.c_double * 3),
        ("normal", C.c_double * 3),
        ("rot", C.c_double * 9),
    ]


class ArtChainReadout(C.Structure):
    _fields_ = [
        ("det", ArtDetectorDesc),
        ("w", C.c_void_p),
        ("cx", C.c_double), ("cy", C.c_double), ("co", C.c_double),
        ("X", C.c_void_p), ("Y", C.c_void_p), ("opl", C.c_void_p),
        ("scratch", C.c_void_p),
        ("out24", C.c_void_p),
        ("lite", C.c_int32),
        ("sums", C.c_int32),
    ]


ART_GUIDES_MAX = 8
ART_XHDR_DOUBLES = 26
ART_ANALYSIS_DOUBLES = 64
ART_JOB_AUTOPLACE, ART_JOB_MANUAL, ART_JOB_SUMS = range(3)


class ArtAnalysisJob(C.Structure):
    _fields_ = [
        ("b", ArtBundleView),
        ("w", C.c_void_p),
        ("distance", C.c_double),
        ("mode", C.c_int32),
        ("reserved", C.c_int32),
        ("centre", C.c_double * 3),
        ("normal", C.c_double * 3),
        ("refpoint", C.c_double * 3),
        ("sums", C.c_void_p),
    ]


# name -> (restype, argtypes); the loader checks every symbol exists (tests/test_abi.py does too)
PROTOTYPES = {
    "art_abi_version": (C.c_int, []),
    "art_last_error": (C.c_char_p, []),
    "art_device_count": (C.c_int, []),
    "art_trace_element": (C.c_int, [C.POINTER(ArtElementDesc), C.POINTER(ArtBundleView), C.POINTER(ArtBundleView),
                                    C.c_int64, C.c_void_p]),
    "art_trace_chain": (C.c_int, [C.POINTER(ArtElementDesc), C.c_int32, C.POINTER(ArtBundleView),
                                  C.POINTER(ArtBundleView), C.c_int64, C.c_void_p]),
    "art_scene_bytes": (C.c_int64, [C.c_int32, C.c_int32]),
    "art_chain_readout_scratch_doubles": (C.c_int64, [C.c_int64]),
    "art_trace_chain_readout": (C.c_int, [C.POINTER(ArtElementDesc), C.c_int32, C.POINTER(ArtBundleView),
                                          C.POINTER(ArtBundleView), C.POINTER(ArtChainReadout), C.c_int64, C.c_void_p]),
    "art_scene_pack": (C.c_int, [C.POINTER(ArtElementDesc), C.c_int32, C.c_int32, C.POINTER(ArtBundleView),
                                 C.POINTER(ArtBundleView), C.POINTER(ArtChainReadout), C.c_void_p]),
    "art_trace_scene": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p]),
    "art_pack_rays": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.POINTER(ArtBundleView), C.c_void_p]),
    "art_transform_bundle": (C.c_int, [c_double_p, c_double_p, C.c_int32, C.POINTER(ArtBundleView),
                                       C.POINTER(ArtBundleView), C.c_int64, C.c_void_p]),
    "art_detector": (C.c_int, [C.POINTER(ArtDetectorDesc), C.POINTER(ArtBundleView), C.c_int64, C.c_void_p,
                               C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "art_detector_readout": (C.c_int, [C.POINTER(ArtDetectorDesc), C.POINTER(ArtBundleView), C.c_void_p, C.c_int64,
                                       C.c_double, C.c_double, C.c_double] + [C.c_void_p] * 9),
    "art_detector_scan_moments": (C.c_int, [C.POINTER(ArtDetectorDesc), C.POINTER(ArtBundleView), C.c_void_p,
                                            C.c_int64, C.c_double, C.c_double, C.c_void_p, C.c_void_p,
                                            C.c_void_p]),
    "art_reduce_scratch_doubles": (C.c_int64, []),
    "art_detector_stats": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64,
                                     C.c_void_p, C.c_void_p, C.c_void_p]),
    "art_detector_moments": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64,
                                       C.c_double, C.c_double, C.c_double, C.c_void_p, C.c_void_p, C.c_void_p]),
    "art_bundle_sums": (C.c_int, [C.POINTER(ArtBundleView), C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p,
                                  C.c_void_p]),
    "art_gaussian_intensity": (C.c_int, [C.POINTER(ArtBundleView), c_double_p, C.c_double, C.c_int64, C.c_void_p,
                                         C.c_void_p, C.c_void_p]),
    "art_gaussian_intensity_central": (C.c_int, [C.POINTER(ArtBundleView), C.c_double, C.c_int64, C.c_void_p, C.c_void_p,
                                                 C.c_void_p, C.c_void_p]),
    "art_bundle_max_angle": (C.c_int, [C.POINTER(ArtBundleView), c_double_p, C.c_int64, C.c_void_p, C.c_void_p,
                                       C.c_void_p]),
    "art_compact_scratch_ints": (C.c_int64, [C.c_int64]),
    "art_compact": (C.c_int, [C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "art_make_source": (C.c_int, [C.c_int32, C.c_double, c_double_p, c_double_p, C.c_int64, C.c_int64, C.c_int64,
                                  C.POINTER(ArtBundleView), C.c_void_p]),
    "art_make_source_strided": (C.c_int, [C.c_int32, C.c_double, c_double_p, c_double_p, C.c_int64, C.c_int64, C.c_int64,
                                          C.c_int64, C.POINTER(ArtBundleView), C.c_void_p]),
    "art_exchange_pack": (C.c_int, [C.c_void_p] * 6 + [C.c_int64, C.c_void_p, C.c_void_p]),
    "art_exchange_fold": (C.c_int, [C.c_void_p, C.c_int32, C.c_int64, C.c_void_p, C.c_void_p]),
    "art_survivor_bytes": (C.c_int64, [C.c_int64, C.c_int32]),
    "art_pack_survivors": (C.c_int, [C.c_void_p, C.c_int64] + [C.c_void_p] * 4 + [C.c_int64, C.c_int64, C.c_void_p,
                                                                                    C.c_void_p, C.c_int64, C.c_void_p]),
    "art_survivor_finish": (C.c_int, [C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p]),
    "art_survivor_xheader": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "art_trace_guides": (C.c_int, [C.POINTER(ArtElementDesc), C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p]),
    "art_analysis_scratch_doubles": (C.c_int64, [C.c_int32, C.c_int64]),
    "art_analyse_bundles": (C.c_int, [C.c_void_p, C.POINTER(ArtAnalysisJob), C.c_int32, C.c_int64, C.c_void_p, C.c_void_p,
                                      C.c_void_p]),
    "art_make_extended_source": (C.c_int, [C.c_double, C.c_double, C.c_int64, C.c_int64, c_double_p, c_double_p,
                                           C.c_int64, C.c_int64, C.POINTER(ArtBundleView), C.c_void_p]),
}


def bind(lib, prefix="art_"):
    """Attach restype/argtypes for every exported entry point; raises AttributeError if one is missing."""
    fns = {}
    for name, (res, args) in PROTOTYPES.items():
        sym = name if prefix == "art_" else prefix + name[len("art_"):]
        f = getattr(lib, sym)
        f.restype = res
        f.argtypes = args
        fns[name] = f
    return fns
