"""Default option dictionaries (values of ART/DefaultOptions.py)."""
DefaultAnalysisOptions = dict(
    verbose=True, plot_Render=False, maxRaysToRender=200, OEPointsToRender=3000, OEPointsScale=5, draw_mesh=False,
    cycle_ray_colors=False, DrawAiryAndFourier=True, plot_SpotDiagram=False, plot_DelaySpotDiagram=False,
    plot_IntensitySpotDiagram=False, plot_IncidenceSpotDiagram=False, plot_DelayGraph=False,
    plot_IntensityGraph=False, plot_IncidenceGraph=False, plot_DelayMirrorProjection=False,
    plot_IntensityMirrorProjection=False, plot_IncidenceMirrorProjection=False, save_results=True)

DefaultSourceProperties = dict(Divergence=0, SourceSize=0, Wavelength=50e-6, DeltaFT=1, NumberRays=1000)

DefaultDetectorOptions = dict(ReflectionNumber=-1, ManualDetector=False, DetectorCentre=None, DetectorNormal=None,
                              DistanceDetector=None, AutoDetectorDistance=False, OptFor="intensity")
