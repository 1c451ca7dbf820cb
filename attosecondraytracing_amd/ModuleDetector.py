"""Plane detector, API of ART/ModuleDetector.py.  The read-out (hit points, 2D coordinates, optical paths)
and its reductions (mean path, bounding box, second moments) run on the GPU for the whole bundle."""
import numpy as np

from . import _abi
from . import ModuleGeometry as mgeo
from . import ModuleProcessing as mp
from .bundle import RayBundle

LightSpeed = 299792458000  # mm/s


def _is_number(x):
    return type(x) in (int, float, np.float64)


class Detector:
    def __init__(self, RefPoint, Centre=None, Normal=None):
        self.centre = Centre
        self.normal = Normal
        self.refpoint = RefPoint

    @property
    def centre(self):
        return self._centre

    @centre.setter
    def centre(self, Centre):
        if Centre is not None and not (isinstance(Centre, np.ndarray) and Centre.shape == (3,)):
            raise TypeError("Detector Centre must be a 3D-vector, given as numpy.ndarray of shape (3,).")
        self._centre = Centre

    @property
    def normal(self):
        return self._normal

    @normal.setter
    def normal(self, Normal):
        if Normal is None:
            self._normal = None
        elif isinstance(Normal, np.ndarray) and Normal.shape == (3,) and mgeo._norm(Normal) > 0:
            self._normal = Normal / mgeo._norm(Normal)
        else:
            raise TypeError("Detector Normal must be a 3D-vector of norm >0, given as numpy.ndarray of shape (3,).")

    @property
    def refpoint(self):
        return self._refpoint

    @refpoint.setter
    def refpoint(self, RefPoint):
        if not (isinstance(RefPoint, np.ndarray) and RefPoint.shape == (3,)):
            raise TypeError("Detector RefPoint must a 3D-vector, given as numpy.ndarray of shape (3,).")
        self._refpoint = RefPoint

    def __getstate__(self):
        """Archive form (mp.save_compressed): the pose only -- not the device analysis attached by autoplace (it holds the
        analysed bundle) nor the cached rotation."""
        st = dict(self.__dict__)
        st.pop("_analysis", None)
        st.pop("_rot_cache", None)
        return st

    # ------------------------------------------------------------------ placement
    def copy_detector(self):
        # the three vectors were validated when this detector got them (and the normal normalised: used bit for bit -- the
        # setter's renormalisation could move its last bit); like the reference's copy, the new detector REFERS to them
        d = Detector.__new__(Detector)
        d._centre, d._normal, d._refpoint = self._centre, self._normal, self._refpoint
        d._analysis = getattr(self, "_analysis", None)
        return d

    def autoplace(self, RayList, DistanceDetector: float):
        """Normal to the central ray of RayList, DistanceDetector away from its origin (ART/ModuleDetector.py:109-137).
        Sums, placement and the read-out moments on the new detector come from ONE device analysis of the bundle
        (analysis.analyse); it stays attached, so that an autofocus search or a result summary that follows on the same
        bundle and detector (ARTmain.run_ART) does not touch the bundle again."""
        from . import analysis
        self._adopt(analysis.analyse([(RayList, "autoplace", DistanceDetector)])[0])

    def _adopt(self, ana):
        """Take the pose a device analysis placed (bit for bit: the setters' normalisation already happened there)."""
        if not ana.count > 0:
            raise TypeError("Detector Normal must be a 3D-vector of norm >0, given as numpy.ndarray of shape (3,).")
        self._normal, self._centre, self._refpoint = ana.normal, ana.centre, ana.refpoint
        self._analysis = ana

    def _analysis_of(self, RayList):
        """The device analysis of `RayList` on THIS detector pose: the attached one if it still applies, else a new one."""
        from . import analysis
        self._iscomplete()
        ana = getattr(self, "_analysis", None)
        if ana is not None and isinstance(RayList, RayBundle) and ana.matches(RayList, (self._centre.tobytes(), self._normal.tobytes())):
            return ana
        ana = analysis.analyse([(RayList, "manual", self)])[0]
        if isinstance(RayList, RayBundle):
            self._analysis = ana
        return ana

    def get_distance(self):
        """ART/ModuleDetector.py:139-145."""
        I = mgeo.IntersectionLinePlane(self.refpoint, -self.normal, self.centre, self.normal)
        return np.float64(mgeo._norm(self.refpoint - I))

    def shiftToDistance(self, NewDistance: float):
        if not _is_number(NewDistance):
            raise TypeError("The new Detector Distance must be int or float.")
        self._centre = self._centre - (NewDistance - self.get_distance()) * self.normal
        self._analysis = None         # (taken at the old pose)

    def shiftByDistance(self, Shift: float):
        if not _is_number(Shift):
            raise TypeError("The Detector Distance Shift must be int or float.")
        self._centre = self.centre - Shift * self.normal
        self._analysis = None         # (taken at the old pose)

    def _iscomplete(self):
        if self.centre is None or self.normal is None:
            raise TypeError("The detector has no centre and normal vectors defined yet.")
        return True

    # ------------------------------------------------------------------ device read-out
    def _desc(self):
        """ArtDetectorDesc; the rotation normal -> ez (ModuleDetector.py:231) is cached until the normal changes."""
        key = self._normal.tobytes()
        cached = getattr(self, "_rot_cache", None)
        if cached is None or cached[0] != key:
            rot = mgeo.rotation_matrix(self.normal, np.array([0.0, 0.0, 1.0])).reshape(9)
            cached = (key, [float(v) for v in rot])
            self._rot_cache = cached
        d = _abi.ArtDetectorDesc()
        d.centre[:] = [float(v) for v in self.centre]
        d.normal[:] = [float(v) for v in self.normal]
        d.rot[:] = cached[1]
        return d

    def _readout_key(self, path_centre=0.0):
        """What a read-out depends on besides the bundle: the detector plane and the provisional path centre."""
        return (self._centre.tobytes(), self._normal.tobytes(), float(path_centre))

    def readout(self, RayList, points3d=False, sync=True, path_centre=0.0, store=True, lite=False):
        """One fused pass on the device (art_detector_readout): per-slot tensors 'X', 'Y' (detector-plane coordinates
        about Detector.centre, ART/ModuleDetector.py:212-234), 'opl' (optical path to the detector, :272-275),
        optionally 'P3' (3-D hit points, :191-210), valid where the bundle is alive, and the 24 statistics
        ('stats': host array; with sync=False 'stats_dev', a device tensor, so that nothing blocks the host).
        `path_centre`: provisional centre for the second moments of the path (see include/art_hip.h);
        store=False skips the per-ray outputs (statistics only); lite=True: the caller needs only count, sum of paths,
        bounding box and path range of the statistics (a LITE fused read-out, ArtChainReadout.lite, may then be used)."""
        self._iscomplete()
        B = RayList if isinstance(RayList, RayBundle) else RayBundle.from_ray_list(RayList)
        fused = getattr(B, "_fused_readout", None)
        if fused is not None and not points3d and fused[0] == self._readout_key(path_centre) and fused[1] == B.version \
                and (fused[2]["X"] is not None or not store) and (lite or not fused[2].get("lite")):
            # computed in the launch that traced this bundle (RayTracingCalculation(..., detector=self))
            res = fused[2]
            if sync and "stats" not in res:
                res["stats"] = res["stats_dev"].cpu().numpy()
            return dict(res, bundle=B)
        be = B.backend
        n = B.n_slots
        X = Y = opl = P3 = None
        if store:
            X, Y, opl = be.empty(n), be.empty(n), be.empty(n)
            P3 = [be.empty(n), be.empty(n), be.empty(n)] if points3d else None
        stats = be.detector_readout(self._desc(), B.view(), B.intensity, n, (0.0, 0.0, path_centre), P3,
                                    (X, Y) if store else None, opl, to_host=sync)
        return {"bundle": B, "X": X, "Y": Y, "opl": opl, "P3": P3, ("stats" if sync else "stats_dev"): stats}

    def _scan_moments(self, RayList, span=0.0):
        """Moment sums from which spot size and duration follow at ANY shift of this detector along its normal
        (art_detector_scan_moments).  Two passes over the bundle: the first only finds the mean path used to centre
        the second."""
        self._iscomplete()
        B = RayList if isinstance(RayList, RayBundle) else RayBundle.from_ray_list(RayList)
        be, n, d = B.backend, B.n_slots, self._desc()
        first = be.detector_scan_moments(d, B.view(), B.intensity, n, 0.0)
        co = first[11] / first[0]
        m = be.detector_scan_moments(d, B.view(), B.intensity, n, co, span)
        # "kinked": some ray's hit changes side of its origin within the span -> its path |t| is not linear there
        return {"m": m, "co": co, "kinked": bool(m[32] > 0)}

    @staticmethod
    def _spot_duration_from_moments(mom, s, weighted):
        """(spot size std, duration std in fs) of the detector shifted by s, from the sums of _scan_moments."""
        m = mom["m"][16:] if weighted else mom["m"][:16]
        var = []
        for k in range(3):
            q, sq, qq, qs, ss = m[1 + 5 * k: 6 + 5 * k]
            mean = (q + s * sq) / m[0]
            var.append(max((qq + 2 * s * qs + s * s * ss) / m[0] - mean * mean, 0.0))
        return float(np.sqrt(var[0] + var[1])), float(np.sqrt(var[2]) / LightSpeed * 1e15)

    # ------------------------------------------------------------------ reference API (host arrays of survivors)
    def get_PointList3D(self, RayList):
        """(m,3) hit points of the surviving rays (ART/ModuleDetector.py:191-210)."""
        r = self.readout(RayList, points3d=True)
        idx = r["bundle"].index()
        return np.stack([p.index_select(0, idx).cpu().numpy() for p in r["P3"]], axis=1)

    def get_PointList2D(self, RayList):
        """(m,2) detector-plane points, origin at Detector.centre (ART/ModuleDetector.py:212-234)."""
        r = self.readout(RayList, lite=True)
        idx = r["bundle"].index()
        return np.stack([r["X"].index_select(0, idx).cpu().numpy(), r["Y"].index_select(0, idx).cpu().numpy()], axis=1)

    def get_PointList2DCentre(self, RayList):
        """(m,2) points centred on their bounding box (ART/ModuleDetector.py:236-252; ModuleGeometry.py:222-245)."""
        r = self.readout(RayList, lite=True)
        s = r["stats"]
        idx = r["bundle"].index()
        cx, cy = (s[3] + s[2]) * 0.5, (s[5] + s[4]) * 0.5
        return np.stack([r["X"].index_select(0, idx).cpu().numpy() - cx,
                         r["Y"].index_select(0, idx).cpu().numpy() - cy], axis=1)

    def get_OpticalPaths(self, RayList):
        r = self.readout(RayList, lite=True)
        return r["opl"].index_select(0, r["bundle"].index()).cpu().numpy()

    def get_Delays(self, RayList):
        """Delays in fs relative to the mean travel time (ART/ModuleDetector.py:254-279)."""
        r = self.readout(RayList, lite=True)
        s = r["stats"]
        opl = r["opl"].index_select(0, r["bundle"].index()).cpu().numpy()
        return (opl - s[1] / s[0]) / LightSpeed * 1e15
