"""matplotlib adaptors for the plot entry points of ART/ModuleAnalysisAndPlots.py (SpotDiagram :133-281, DelayGraph
:284-441, MirrorProjection :444-525, RayRenderGraph :529-673), fed from device-resident bundles.

Numbers shown in the legends (spot size, standard deviations, numerical aperture) are reduced on the device over
ALL rays; the scatter itself shows at most MAX_POINTS survivors, evenly spaced in ray order, because a figure cannot
resolve more and 1e7 markers would take minutes to draw.  matplotlib is imported on first use only, so the tracing
path never depends on it.  The 3-D scene render (RayRenderGraph) is PyVista's in the reference; the image has no
PyVista, so the same geometry (`render_scene`) is drawn on matplotlib's 3-D axes."""
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


def _same_slots(a, b):
    """Slot i of both bundles is the same source ray (bundles of one trace share their `number` tensor)."""
    return a.n_slots == b.n_slots and a.number is b.number


def _ray_segments(history, EndDistance, maxRays):
    """Per stage k of the history a (2 m, 3) array of segment end points, pairs in ray order: from the ray's point in
    bundle k to its point in bundle k + 1 for the rays that are still alive there; for the last bundle, from the point
    along the direction over EndDistance (ART/ModuleAnalysisAndPlots.py:563-602).  At most maxRays rays per stage --
    evenly spaced in ray order here, a random draw in the reference.  Only the drawn rays leave the device."""
    import torch
    out = []
    for k, B in enumerate(history):
        last = k == len(history) - 1
        nxt = B if last else history[k + 1]
        pos = _sample_positions(len(nxt), cap=maxRays)
        slots = nxt.index().index_select(0, torch.as_tensor(pos, device=nxt.backend.device))
        P2 = nxt.data[0:3].index_select(1, slots).cpu().numpy().T
        if last:
            P1, P2 = P2, P2 + nxt.data[3:6].index_select(1, slots).cpu().numpy().T * EndDistance
        elif _same_slots(B, nxt):
            P1 = B.data[0:3].index_select(1, slots).cpu().numpy().T
        else:       # bundles that do not share their slots (e.g. built from Ray lists): match the ray numbers
            mine = B.numbers()
            order = np.argsort(mine, kind="stable")
            want = nxt.numbers()[pos]
            at = order[np.searchsorted(mine, want, sorter=order)]
            if not np.array_equal(mine[at], want):
                raise ValueError("a ray of bundle %d has no ancestor in bundle %d" % (k + 1, k))
            P1 = B.points()[at]
        seg = np.empty((2 * len(pos), 3))
        seg[0::2], seg[1::2] = P1, P2
        out.append(seg)
    return out


def _optic_cloud(OE, OEpoints, draw_mesh=False):
    """Sample points of one optical element's surface in the lab frame and the closed index loops of its hole outlines
    (ART/ModuleAnalysisAndPlots.py:529-561).  Without a mesh the cloud sits 0.5 mm behind the surface, so that the ray
    ends on it stay visible."""
    from . import ModuleGeometry as mgeo
    pts, loops = OE.type.get_grid3D(OEpoints, edges=True)
    P = np.asarray(pts, dtype=float).reshape(-1, 3) - np.asarray(OE.type.get_centre(), dtype=float)
    _, bwd = mgeo.frame_maps(OE.normal, OE.majoraxis)
    P = P @ bwd.T + np.asarray(OE.position, dtype=float)
    if not draw_mesh:
        P = P - 0.5 * np.asarray(OE.normal, dtype=float)
    return P, loops


def _optic_triangles(OE, OEpoints):
    """Triangles (index triples into the cloud of _optic_cloud) of the optic's surface for draw_mesh=True: a Delaunay
    triangulation of the sample points in the plane of the support, without the triangles that bridge a hole or a
    concave outline (the reference asks PyVista for a Delaunay mesh constrained by the hole outlines, :544-560)."""
    from matplotlib.tri import Triangulation
    pts = np.asarray(OE.type.get_grid3D(OEpoints), dtype=float).reshape(-1, 3)
    xy = pts[:, :2] - np.asarray(OE.type.get_centre(), dtype=float)[:2]
    if len(xy) < 3:
        return np.empty((0, 3), dtype=np.int64)
    tri = Triangulation(xy[:, 0], xy[:, 1]).triangles
    mid = xy[tri].mean(axis=1)
    inside = np.fromiter((bool(OE.type.support._IncludeSupport(m)) for m in mid), dtype=bool, count=len(mid))
    return tri[inside].astype(np.int64)


def render_scene(OpticalChain, EndDistance=None, maxRays=300, OEpoints=3000, draw_mesh=False):
    """The geometry RayRenderGraph draws, as host arrays (for any renderer): {"segments": one (2 m, 3) array of
    segment end points per stage (source -> element 0, ..., last element -> EndDistance further), "optics": one
    (p, 3) lab-frame point cloud per optical element, "loops": their hole outlines as index loops, "EndDistance",
    "triangles": with draw_mesh, the surface mesh of every optic as index triples into its cloud}."""
    history = [_as_bundle(OpticalChain.source_rays)] + [_as_bundle(b) for b in OpticalChain.get_output_rays()]
    if EndDistance is None:
        EndDistance = float(np.linalg.norm(np.asarray(OpticalChain.source_rays[0].point, dtype=float)
                                           - np.asarray(OpticalChain.optical_elements[0].position, dtype=float)))
    clouds = [_optic_cloud(OE, OEpoints, draw_mesh) for OE in OpticalChain.optical_elements]
    return {"segments": _ray_segments(history, EndDistance, maxRays), "optics": [c[0] for c in clouds],
            "loops": [c[1] for c in clouds], "EndDistance": EndDistance,
            "triangles": [_optic_triangles(OE, OEpoints) for OE in OpticalChain.optical_elements] if draw_mesh else None}


def generate_distinct_colors(num_colors):
    """num_colors visually distinct colours (the reference takes colorcet's glasbey palette, :604-614; matplotlib's
    tab20 here)."""
    cmap = _plt().get_cmap("tab20")
    return [cmap(i % 20)[:3] for i in range(num_colors)]


def RayRenderGraph(OpticalChain, EndDistance=None, maxRays=300, OEpoints=3000, scale_spheres=5.0, draw_mesh=False,
                   cycle_ray_colors=False):
    """3-D picture of the optical setup and of at most maxRays traced rays (ART/ModuleAnalysisAndPlots.py:616-673), on
    matplotlib's 3-D axes.  Returns the figure; `fig._art_scene` holds the arrays that were drawn (render_scene)."""
    import colorsys
    from mpl_toolkits.mplot3d.art3d import Line3DCollection
    plt = _plt()
    print("...rendering image of optical chain...", end="", flush=True)
    scene = render_scene(OpticalChain, EndDistance, maxRays, OEpoints, draw_mesh)
    n_stage = len(scene["segments"])
    colors = generate_distinct_colors(n_stage) if cycle_ray_colors else [(0.7, 0.0, 0.0)] * n_stage
    fig = plt.figure(figsize=(15, 5))
    ax = fig.add_subplot(111, projection="3d")
    ax.view_init(elev=20, azim=-75)
    ax.set_proj_type("ortho")           # (a perspective camera clips the near end of a zoomed, elongated box)
    for seg, color in zip(scene["segments"], colors):
        ax.add_collection3d(Line3DCollection(seg.reshape(-1, 2, 3), colors=[color], linewidths=0.6))
    everything = [seg for seg in scene["segments"] if len(seg)]
    for i, (cloud, loops) in enumerate(zip(scene["optics"], scene["loops"])):
        h, sat, v = colorsys.rgb_to_hsv(*colors[min(i + 1, n_stage - 1)])
        pale = colorsys.hsv_to_rgb(h, 0.2 * sat, v)        # the optic in the pale shade of the rays that leave it
        ax.scatter(cloud[:, 0], cloud[:, 1], cloud[:, 2], s=scale_spheres, color=[pale], depthshade=False)
        if draw_mesh:
            tri = scene["triangles"][i]
            if len(tri):
                ax.plot_trisurf(cloud[:, 0], cloud[:, 1], cloud[:, 2], triangles=tri, color=pale, alpha=0.6, linewidth=0)
            for loop in loops:
                ax.plot(*cloud[loop].T, color=pale, linewidth=1.0)
        everything.append(cloud)
    lo, hi = np.concatenate(everything).min(axis=0), np.concatenate(everything).max(axis=0)
    span = np.maximum(hi - lo, 0.08 * float((hi - lo).max()))      # a beam line is thin: no axis flatter than 8 % of the longest
    mid = 0.5 * (lo + hi)
    for setter, c, w in zip((ax.set_xlim, ax.set_ylim, ax.set_zlim), mid, span):
        setter(c - 0.5 * w, c + 0.5 * w)
    ax.set_box_aspect(tuple(span), zoom=1.9)               # equal scales on the three axes, the box shaped like the setup
    fig.subplots_adjust(left=0.0, right=1.0, bottom=0.0, top=1.0)
    ax.set_xlabel("x (mm)")
    ax.set_ylabel("y (mm)")
    ax.set_zlabel("z (mm)")
    fig._art_scene = scene
    print("\r\033[K", end="", flush=True)
    return fig


def show():
    plt = _plt()
    plt.show(block=False)
