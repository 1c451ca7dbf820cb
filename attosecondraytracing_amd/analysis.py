"""Analysis of traced bundles on the device, for MANY bundles at once (art_analyse_bundles, include/art_hip.h).

What the reference's launcher computes per chain of a loop list -- energy transmission (ART/ModuleAnalysisAndPlots.py:62-77),
detector auto-placement on the mean ray (ART/ModuleDetector.py:109-137), the autofocus search (ART/ModuleProcessing.py:
317-460) or the result summary (ART/ModuleAnalysisAndPlots.py:81-129) -- needs, per bundle, a handful of global sums and
ONE set of read-out moments: the read-out of every ray is linear in a shift of the detector along its normal, so spot
size and duration at every position of every scan follow from the same 32 sums.  `analyse()` gets them for a whole list
of bundles in four launches (blockIdx.y = bundle) and ONE device-to-host copy; everything after that is arithmetic on 64
doubles per bundle."""
import numpy as np

from . import _abi
from .bundle import RayBundle

LightSpeed = 299792458000  # mm/s
MAX_JOBS_PER_CALL = 64


class BundleAnalysis:
    """One row of art_analyse_bundles' output on the host (layout: include/art_hip.h)."""

    def __init__(self, row, bundle, mode):
        # the analysed bundle is remembered by IDENTITY (serial number + version), not held: a Detector keeps its analysis,
        # and a detector kept in `kept_data` must not pin 650 MB of device memory per 1e7 rays for its lifetime
        self.row, self.mode = row, mode
        self._serial, self.version = bundle._serial, bundle.version

    count = property(lambda self: self.row[0])
    sum_w = property(lambda self: self.row[7])
    centre = property(lambda self: self.row[10:13].copy())
    normal = property(lambda self: self.row[13:16].copy())
    refpoint = property(lambda self: self.row[16:19].copy())
    co = property(lambda self: self.row[19])
    moments = property(lambda self: self.row[20:53])
    kink_below = property(lambda self: self.row[53])      # largest shift <= 0 at which some ray's path has its kink
    kink_above = property(lambda self: self.row[54])      # smallest shift > 0
    max_angle = property(lambda self: self.row[55])
    bbox = property(lambda self: self.row[56:62])         # min X, max X, min Y, max Y, min opl, max opl at shift 0

    def mean_point(self):
        return self.row[1:4] / self.row[0]

    def mean_vector(self):
        return self.row[4:7] / self.row[0]

    def spot_duration(self, s, weighted):
        """(spot size std in mm, duration std in fs) of the detector shifted by s along -normal."""
        return spot_duration_from_moments(self.moments, s, weighted)

    def linear_over(self, lo, hi):
        """Is every ray's read-out linear for all shifts in [min(lo, 0), max(hi, 0)]?  (No hit point passes through its
        ray's origin there; conservative: only the two kinks nearest to shift 0 are known.)"""
        return lo > self.kink_below and hi < self.kink_above

    def matches(self, bundle, detector_key=None):
        return (self._serial == bundle._serial and self.version == bundle.version
                and (detector_key is None or detector_key == self.key()))

    def key(self):
        return (self.row[10:13].tobytes(), self.row[13:16].tobytes())


def spot_duration_from_moments(m, s, weighted):
    """m: the 32 (+1) sums of art_detector_scan_moments / art_analyse_bundles."""
    m = m[16:] if weighted else m[:16]
    var = []
    for k in range(3):
        q, sq, qq, qs, ss = m[1 + 5 * k: 6 + 5 * k]
        mean = (q + s * sq) / m[0]
        var.append(max((qq + 2 * s * qs + s * s * ss) / m[0] - mean * mean, 0.0))
    return float(np.sqrt(var[0] + var[1])), float(np.sqrt(var[2]) / LightSpeed * 1e15)


def _job(bundle, mode, arg):
    j = _abi.ArtAnalysisJob()
    j.b = bundle.view()
    j.w = None if bundle.intensity is None else bundle.intensity.data_ptr()
    fs = bundle.fused_sums()          # pass (1) formed by the tracing launch: the analysis reads the bundle once
    j.sums = None if fs is None else fs.data_ptr()
    j.mode = mode
    if mode == _abi.ART_JOB_AUTOPLACE:
        j.distance = float(arg)
    elif mode == _abi.ART_JOB_MANUAL:
        arg._iscomplete()
        j.centre[:] = [float(v) for v in arg.centre]
        j.normal[:] = [float(v) for v in arg.normal]
        j.refpoint[:] = [float(v) for v in arg.refpoint]
    return j


def analyse(requests):
    """requests: [(bundle, "sums" | "autoplace" | "manual", argument)] -- argument: None / DistanceDetector / a placed
    Detector.  Returns one BundleAnalysis per request, in order.  Bundles of equal slot count share one call of
    art_analyse_bundles; all calls are enqueued before the first result is read: one host synchronisation in total."""
    modes = {"sums": _abi.ART_JOB_SUMS, "autoplace": _abi.ART_JOB_AUTOPLACE, "manual": _abi.ART_JOB_MANUAL}
    groups = {}
    for pos, (bundle, kind, arg) in enumerate(requests):
        B = bundle if isinstance(bundle, RayBundle) else RayBundle.from_ray_list(bundle)
        groups.setdefault((id(B.backend), B.n_slots), []).append((pos, B, modes[kind], arg))
    pending = []
    for (_, n), items in groups.items():
        be = items[0][1].backend
        if n == 0:
            pending.append((items, None))
            continue
        for lo in range(0, len(items), MAX_JOBS_PER_CALL):      # (bounds the scratch area: 0.4 MB per job)
            part = items[lo:lo + MAX_JOBS_PER_CALL]
            jobs = [_job(B, mode, arg) for _, B, mode, arg in part]
            pending.append((part, be.analyse_bundles(jobs, n)))
    results = [None] * len(requests)
    for items, out in pending:
        if out is None:
            rows = np.zeros((len(items), _abi.ART_ANALYSIS_DOUBLES))
            rows[:, 10:20] = np.nan
        else:
            rows = out.cpu().numpy()                     # the ONE copy (per group) back to the host
        for (pos, B, mode, _), row in zip(items, rows):
            results[pos] = BundleAnalysis(row, B, mode)
    return results
