"""matplotlib adaptors for the plot entry points of ART/ModuleAnalysisAndPlots.py (SpotDiagram :133-281, DelayGraph
:284-441, MirrorProjection :444-525), fed from device-resident bundles.

Numbers shown in the legends (spot size, standard deviations, numerical aperture) are reduced on the device over
ALL rays; the scatter itself shows at most MAX_POINTS survivors, evenly spaced in ray order, because a figure cannot
resolve more and 1e7 markers would take minutes to draw.  matplotlib is imported on first use only, so the tracing
path never depends on it.  The 3-D scene render of the reference (RayRenderGraph) needs PyVista and is not built."""
import numpy as np

from . import ModuleProcessing as mp
from .bundle import RayBundle

MAX_POINTS = 20000
_COLOUR_LABELS = {"Intensity": "Intensity (arb.u.)", "Incidence": "Incidence angle (deg)", "Delay": "Delay (fs)"}


def _plt():
    import matplotlib.pyplot as plt
    return plt


def _as_bundle(rays):
    return rays if isinstance(rays, RayBundle) else RayBundle.from_ray_list(rays)


def _sample_positions(m, cap=None):
    cap = cap or MAX_POINTS
    return np.arange(m) if m <= cap else np.unique(np.linspace(0, m - 1, cap).astype(np.int64))


def _detector_sample(B, Detector, pos):
    """Read-out of the bundle on the detector: per-ray values of the sampled survivors + statistics of all rays."""
    import torch
    from .ModuleDetector import LightSpeed
    ro = Detector.readout(B, sync=True)
    s = ro["stats"]
    slots = B.index().index_select(0, torch.as_tensor(pos, device=B.backend.device))
    take = lambda t: t.index_select(0, slots).cpu().numpy()
    cx, cy = 0.5 * (s[2] + s[3]), 0.5 * (s[4] + s[5])          # CentrePointList: bounding-box centre
    mean_opl = s[1] / s[0]
    spot_sd, dur_sd = Detector._spot_duration_from_moments(Detector._scan_moments(B), 0.0, False)
    return {"x_um": (take(ro["X"]) - cx) * 1e3, "y_um": (take(ro["Y"]) - cy) * 1e3,
            "delay_fs": (take(ro["opl"]) - mean_opl) / LightSpeed * 1e15,
            "size": max(s[3] - s[2], s[5] - s[4]), "spot_sd": spot_sd, "dur_sd": dur_sd, "slots": slots}


def _ray_property(B, slots, which):
    if which == "Intensity":
        if B.intensity is None:
            raise TypeError("rays carry no intensity")
        return B.intensity.index_select(0, slots).cpu().numpy()
    if which == "Incidence":
        return np.rad2deg(B.data[7].index_select(0, slots).cpu().numpy())
    raise ValueError(which)


def _getDetectorPoints(RayListAnalysed, Detector):
    """(x in um, y in um, spot diameter in mm, spot standard deviation in mm), ART/ModuleAnalysisAndPlots.py:28-58;
    the coordinate arrays hold the displayed sample, the two numbers describe all rays."""
    B = _as_bundle(RayListAnalysed)
    d = _detector_sample(B, Detector, _sample_positions(len(B)))
    return d["x_um"], d["y_um"], d["size"], d["spot_sd"]


def _dist_step(size, NA):
    return min(50, max(0.0005, round(size / 8 / np.arcsin(NA) * 10000) / 10000))


def _shift(moving, dist, step, key):
    """Cursor-key handling shared by the interactive figures: returns the new distance or None."""
    if key == "right":
        moving.shiftByDistance(step)
        return dist + step
    if key == "left":
        if dist > 1.5 * step:
            moving.shiftByDistance(-step)
            return dist - step
        moving.shiftToDistance(0.5 * step)
        return 0.5 * step
    return None


def SpotDiagram(RayListAnalysed, Detector, DrawAiryAndFourier=False, ColorCoded=None):
    plt = _plt()
    B = _as_bundle(RayListAnalysed)
    NA = mp.ReturnNumericalAperture(B, 1)
    airy = mp.ReturnAiryRadius(B.wavelength, NA) * 1e3 if DrawAiryAndFourier else 0
    pos = _sample_positions(len(B))
    state = {"dist": Detector.get_distance(), "det": Detector.copy_detector()}

    def colours(d):
        if ColorCoded == "Delay":
            return d["delay_fs"]
        if ColorCoded in ("Intensity", "Incidence"):
            return _ray_property(B, d["slots"], ColorCoded)
        return "red"

    def label(d):
        extra = "\n{:.2f} fs SD".format(d["dur_sd"]) if ColorCoded == "Delay" else ""
        return "{:.3f} mm\n{:.1f} μm SD".format(state["dist"], d["spot_sd"] * 1e3) + extra

    d = _detector_sample(B, Detector, pos)
    state["step"] = _dist_step(d["size"], NA)
    plt.ion()
    fig, ax = plt.subplots()
    if DrawAiryAndFourier:
        th = np.linspace(0, 2 * np.pi, 100)
        ax.plot(airy * np.cos(th), airy * np.sin(th), c="black")
    sc = ax.scatter(d["x_um"], d["y_um"], c=colours(d), s=15, label=label(d))
    cbar = None
    if ColorCoded in _COLOUR_LABELS:
        cbar = fig.colorbar(sc)
        cbar.set_label(_COLOUR_LABELS[ColorCoded])
    head = {None: "Spot Diagram", "Intensity": "Intensity + Spot Diagram", "Incidence": "Ray Incidence + Spot Diagram",
            "Delay": "Delay + Spot Diagram"}.get(ColorCoded, "Spot Diagram")
    ax.set_title(head + "\n press left/right to move detector position")
    ax.set_xlabel("X (µm)")
    ax.set_ylabel("Y (µm)")

    def frame(d):
        lim = 1.1 * max(airy, 0.5 * d["size"] * 1000)
        ax.set_xlim(-lim, lim)
        ax.set_ylim(-lim, lim)
        ax.legend(loc="upper right")

    frame(d)

    def press(event):
        new = _shift(state["det"], state["dist"], state["step"], event.key)
        if new is None:
            return
        state["dist"] = new
        d = _detector_sample(B, state["det"], pos)
        sc.set_offsets(np.column_stack([d["x_um"], d["y_um"]]))
        if ColorCoded == "Delay":
            sc.set_array(d["delay_fs"])
            sc.set_clim(d["delay_fs"].min(), d["delay_fs"].max())
            cbar.update_normal(sc)
        sc.set_label(label(d))
        frame(d)
        state["step"] = _dist_step(d["size"], NA)
        fig.canvas.draw_idle()

    fig.canvas.mpl_connect("key_press_event", press)
    fig._art_press = press          # lets tests drive the handler without a GUI event loop
    plt.show()
    return fig


def _draw_delay_graph(B, Detector, dist, DeltaFT, DrawAiryAndFourier, ColorCoded, fig, pos, NA):
    plt = _plt()
    airy = mp.ReturnAiryRadius(B.wavelength, NA) * 1e3
    d = _detector_sample(B, Detector, pos)
    if fig is None:
        fig = plt.figure()
    else:
        fig.clear(keep_observers=True)
    ax = fig.add_subplot(111, projection="3d")
    ax.set_xlabel("X (µm)")
    ax.set_ylabel("Y (µm)")
    ax.set_zlabel("Delay (fs)")
    lab = "{:.3f} mm\n{:.1f} μm SD\n{:.2f} fs SD".format(dist, d["spot_sd"] * 1e3, d["dur_sd"])
    c = _ray_property(B, d["slots"], ColorCoded) if ColorCoded in ("Intensity", "Incidence") else d["delay_fs"]
    ax.scatter(d["x_um"], d["y_um"], d["delay_fs"], s=4, c=c, label=lab)
    ax.set_title({"Intensity": "Delay + Intensity graph", "Incidence": "Delay + Incidence graph"}.get(ColorCoded, "Delay graph")
                 + "\n press left/right to move detector position")
    ax.legend(loc="upper right")
    if DrawAiryAndFourier:
        x = np.linspace(-airy, airy, 40)
        z = np.linspace(d["delay_fs"].mean() - DeltaFT * 0.5, d["delay_fs"].mean() + DeltaFT * 0.5, 40)
        x, z = np.meshgrid(x, z)
        y = np.sqrt(np.maximum(airy ** 2 - x ** 2, 0.0))
        ax.plot_wireframe(x, y, z, color="grey", alpha=0.1)
        ax.plot_wireframe(x, -y, z, color="grey", alpha=0.1)
    lim = 1.1 * max(airy, 0.5 * d["size"] * 1000)
    ax.set_xlim(-lim, lim)
    ax.set_ylim(-lim, lim)
    return fig, d["size"]


def DelayGraph(RayListAnalysed, Detector, DeltaFT, DrawAiryAndFourier=False, ColorCoded=None):
    plt = _plt()
    B = _as_bundle(RayListAnalysed)
    NA = mp.ReturnNumericalAperture(B, 1)
    pos = _sample_positions(len(B))
    state = {"dist": Detector.get_distance(), "det": Detector.copy_detector()}
    plt.ion()
    fig, size = _draw_delay_graph(B, Detector, state["dist"], DeltaFT, DrawAiryAndFourier, ColorCoded, None, pos, NA)
    state["step"] = _dist_step(size, NA)

    def press(event):
        new = _shift(state["det"], state["dist"], state["step"], event.key)
        if new is None:
            return
        state["dist"] = new
        ax = fig.axes[0]
        view = (ax.azim, ax.elev)
        _, size = _draw_delay_graph(B, state["det"], new, DeltaFT, DrawAiryAndFourier, ColorCoded, fig, pos, NA)
        fig.axes[0].view_init(elev=view[1], azim=view[0])
        state["step"] = _dist_step(size, NA)
        fig.canvas.draw_idle()

    fig.canvas.mpl_connect("key_press_event", press)
    fig._art_press = press
    plt.show()
    return fig


def MirrorProjection(OpticalChain, ReflectionNumber: int, Detector=None, ColorCoded=None):
    plt = _plt()
    import torch
    from mpl_toolkits.axes_grid1 import make_axes_locatable
    from . import ModuleGeometry as mgeo
    oe = OpticalChain.optical_elements[ReflectionNumber]
    B = OpticalChain.get_output_rays()[ReflectionNumber]
    pos = _sample_positions(len(B))
    slots = B.index().index_select(0, torch.as_tensor(pos, device=B.backend.device))
    # hit points in the support frame: the optic frame without the shift to the mirror centre
    fwd, _ = mgeo.frame_maps(oe.normal, oe.majoraxis)
    P = B.data[0:3].index_select(1, slots).cpu().numpy().T - np.asarray(oe.position, dtype=float)
    xy = P @ fwd.T
    if ColorCoded in ("Intensity", "Incidence"):
        z = _ray_property(B, slots, ColorCoded)
    elif ColorCoded == "Delay":
        if Detector is None:
            raise ValueError("If you want to project ray delays, you must specify a detector.")
        z = _detector_sample(B, Detector, pos)["delay_fs"]
    else:
        z = "red"
    title = {"Intensity": "Ray intensity projected on mirror              ",
             "Incidence": "Ray incidence projected on mirror              ",
             "Delay": "Ray delay at detector projected on mirror              "}.get(ColorCoded, "Ray impact points projected on mirror")
    plt.ion()
    fig = plt.figure()
    ax = oe.type.support._ContourSupport(fig)
    p = ax.scatter(xy[:, 0], xy[:, 1], c=z, s=15)
    if ColorCoded in _COLOUR_LABELS:
        cax = make_axes_locatable(ax).append_axes("right", size="5%", pad=0.05)
        fig.colorbar(p, cax=cax).set_label(_COLOUR_LABELS[ColorCoded])
    ax.set_xlabel("x (mm)")
    ax.set_ylabel("y (mm)")
    ax.set_title(title, loc="right")
    ax.autoscale_view()
    fig.tight_layout()
    plt.show()
    return fig


def show():
    plt = _plt()
    plt.show(block=False)
