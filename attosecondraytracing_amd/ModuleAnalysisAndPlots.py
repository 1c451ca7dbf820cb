"""Result summaries and plots, API of ART/ModuleAnalysisAndPlots.py.

`getETransmission` and `GetResultSummary` (the two functions ARTmain needs for its numbers) are built on the
device reductions.  SpotDiagram, DelayGraph and MirrorProjection are matplotlib adaptors fed from the device
(_plots.py: statistics over all rays, markers for a down-sampled subset); RayRenderGraph, a PyVista scene in the
reference, is drawn on matplotlib's 3-D axes from the same geometry (the image has no PyVista)."""

from . import ModuleGeometry as mgeo
from . import ModuleProcessing as mp
from .bundle import RayBundle

def _sum_intensity(rays):
    if isinstance(rays, RayBundle):
        if rays.intensity is None:
            raise TypeError("rays carry no intensity")
        # the sum of a bundle's intensities is remembered with the bundle (per version): the chains of a loop list share
        # their source, and ARTmain asks for its sum once per chain
        hit = getattr(rays, "_sum_w", None)
        if hit is None or hit[0] != rays.version:
            hit = rays._sum_w = (rays.version, float(rays.backend.bundle_sums(rays.view(), rays.intensity, rays.n_slots)[7]))
        return hit[1]
    return sum(r.intensity for r in rays)


def getETransmission(RayListIn, RayListOut) -> float:
    """Energy transmission in percent (ART/ModuleAnalysisAndPlots.py:62-77)."""
    return 100 * _sum_intensity(RayListOut) / _sum_intensity(RayListIn)


def _summary_from_analysis(Detector, ana, verbose=False):
    """GetResultSummary's numbers from a device analysis of the bundle on `Detector` (analysis.BundleAnalysis)."""
    from .ModuleDetector import LightSpeed
    FocalSpotSizeSD, DurationSD = ana.spot_duration(0.0, False)
    if verbose:
        s = ana.bbox
        FocalSpotSize = max(s[1] - s[0], s[3] - s[2])
        delay_range = (s[5] - s[4]) / LightSpeed * 1e15
        print("At the detector distance of " + "{:.3f}".format(Detector.get_distance()) + " mm we get:\n"
              + "Spatial std : " + "{:.3f}".format(FocalSpotSizeSD * 1e3) + " μm and min-max: "
              + "{:.3f}".format(FocalSpotSize * 1e3) + " μm\n"
              + "Temporal std : " + "{:.3e}".format(DurationSD) + " fs and min-max : "
              + "{:.3e}".format(delay_range) + " fs")
    return FocalSpotSizeSD, DurationSD


def GetResultSummary(Detector, RayListAnalysed, verbose=False):
    """Spot-size and duration standard deviations at the detector (ART/ModuleAnalysisAndPlots.py:81-129).
    For a RayBundle everything is reduced on the device (analysis.analyse: moment sums + bounding box in one pass, reused
    if `Detector.autoplace` has just analysed this bundle); no per-ray array is copied."""
    if isinstance(RayListAnalysed, RayBundle):
        return _summary_from_analysis(Detector, Detector._analysis_of(RayListAnalysed), verbose)
    P = Detector.get_PointList2DCentre(RayListAnalysed)
    FocalSpotSizeSD = mp.StandardDeviation(P)
    DelayList = Detector.get_Delays(RayListAnalysed)
    DurationSD = mp.StandardDeviation(DelayList)
    if verbose:
        FocalSpotSize = mgeo.DiameterPointList(P)
        delay_range = max(DelayList) - min(DelayList)
        print("At the detector distance of " + "{:.3f}".format(Detector.get_distance()) + " mm we get:\n"
              + "Spatial std : " + "{:.3f}".format(FocalSpotSizeSD * 1e3) + " μm and min-max: "
              + "{:.3f}".format(FocalSpotSize * 1e3) + " μm\n"
              + "Temporal std : " + "{:.3e}".format(DurationSD) + " fs and min-max : "
              + "{:.3e}".format(delay_range) + " fs")
    return FocalSpotSizeSD, DurationSD


def _getDetectorPoints(RayListAnalysed, Detector):
    from . import _plots
    return _plots._getDetectorPoints(RayListAnalysed, Detector)


def SpotDiagram(RayListAnalysed, Detector, DrawAiryAndFourier=False, ColorCoded=None):
    """Spot diagram on the detector, optionally colour-coded by "Intensity", "Incidence" or "Delay"; left/right keys
    move the detector (ART/ModuleAnalysisAndPlots.py:133-281)."""
    from . import _plots
    return _plots.SpotDiagram(RayListAnalysed, Detector, DrawAiryAndFourier, ColorCoded)


def DelayGraph(RayListAnalysed, Detector, DeltaFT, DrawAiryAndFourier=False, ColorCoded=None):
    """3-D spot diagram with the ray delays on the third axis (ART/ModuleAnalysisAndPlots.py:360-441)."""
    from . import _plots
    return _plots.DelayGraph(RayListAnalysed, Detector, DeltaFT, DrawAiryAndFourier, ColorCoded)


def MirrorProjection(OpticalChain, ReflectionNumber: int, Detector=None, ColorCoded=None):
    """Impact points on one optical element in its support frame (ART/ModuleAnalysisAndPlots.py:444-525)."""
    from . import _plots
    return _plots.MirrorProjection(OpticalChain, ReflectionNumber, Detector, ColorCoded)


def RayRenderGraph(OpticalChain, EndDistance=None, maxRays=300, OEpoints=3000, scale_spheres=5.0, draw_mesh=False,
                   cycle_ray_colors=False):
    """3-D picture of the optical setup and the traced rays (ART/ModuleAnalysisAndPlots.py:616-673)."""
    from . import _plots
    return _plots.RayRenderGraph(OpticalChain, EndDistance, maxRays, OEpoints, scale_spheres, draw_mesh, cycle_ray_colors)


def generate_distinct_colors(num_colors):
    from . import _plots
    return _plots.generate_distinct_colors(num_colors)


def show():
    from . import _plots
    return _plots.show()
