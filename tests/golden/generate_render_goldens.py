#!/usr/bin/env python3
"""Golden vectors of the reference's 3-D scene render (`RayRenderGraph`, ART/ModuleAnalysisAndPlots.py:529-673), made by
RUNNING THE REFERENCE in the build container, like generate_goldens.py (same stand-ins, same rules: TEST
INFRASTRUCTURE, nothing on the GPU box imports this script or the reference).

What executes: the reference's `Support._get_grid` / `_Contour_points`, `get_grid3D` of every optic class, `_RenderRays`
and `_RenderOpticalElement`.  The image has no PyVista: the two calls those functions make into it
(`pv.line_segments_from_points`, `pv.PolyData`) are given pass-through stand-ins that return the point arrays they
receive -- the fixture pins the GEOMETRY handed to the renderer, not the renderer.  `_RenderRays` draws a random
subset when a bundle has more than `maxRays` rays; the fixtures use maxRays above the bundle size (no random draw).

    python tests/golden/generate_render_goldens.py
"""
import json
import os

import numpy as np

import generate_goldens as gg          # sets up sys.path (reference + stand-ins) and imports the reference modules

mplots, mmirror, mmask, msupp, mp = gg.mplots, gg.mmirror, gg.mmask, gg.msupp, gg.mp

mplots.pv.line_segments_from_points = lambda points: np.asarray(points, dtype=float)
mplots.pv.PolyData = lambda points, **kw: np.asarray(points, dtype=float)


def supports():
    return {"round": msupp.SupportRound(12.0), "roundhole": msupp.SupportRoundHole(30, 5, 10, 5),
            "rect": msupp.SupportRectangle(150, 32), "recthole": msupp.SupportRectangleHole(60, 40, 6, 8, -5),
            "rectrecthole": msupp.SupportRectangleRectHole(60, 40, 12, 8, 10, 4)}


def optics():
    R, r = mmirror.ReturnOptimalToroidalRadii(600, 80)
    return {"plane": mmirror.MirrorPlane(msupp.SupportRectangleRectHole(60, 40, 12, 8, 10, 4)),
            "sphere": mmirror.MirrorSpherical(400.0, msupp.SupportRound(25)),
            "parabola": mmirror.MirrorParabolic(100, 90, msupp.SupportRoundHole(30, 5, 10, 5)),
            "torus": mmirror.MirrorToroidal(R, r, msupp.SupportRectangle(200, 30)),
            "ellipsoid": mmirror.MirrorEllipsoidal(msupp.SupportRectangle(60, 30), SemiMajorAxis=500, SemiMinorAxis=80,
                                                   OffAxisAngle=30),
            "cylinder": mmirror.MirrorCylindrical(300.0, msupp.SupportRectangleHole(60, 40, 6, 8, -5)),
            "mask": mmask.Mask(msupp.SupportRoundHole(30, 10.25, 0, 0))}


def main():
    arrays, scene = {}, {"supports": {}, "optics": {}}
    for name, S in supports().items():
        scene["supports"][name] = gg.describe_support(S)
        for n in (200, 37):
            arrays[f"sup_{name}_grid{n}"] = np.array(S._get_grid(n), dtype=float).reshape(-1, 2)
        pts, edges = S._Contour_points(40, edges=True)
        arrays[f"sup_{name}_contour40"] = np.array(pts, dtype=float).reshape(-1, 2)
        scene["supports"][name]["contour40_edges"] = [[int(i) for i in e] for e in edges]
    for name, O in optics().items():
        d = gg.describe_optic(O, arrays, f"opt_{name}_")
        pts, edges = O.get_grid3D(300, edges=True)
        arrays[f"opt_{name}_grid300"] = np.array(pts, dtype=float).reshape(-1, 3)
        d["grid300_edges"] = [[int(i) for i in e] for e in edges]
        arrays[f"opt_{name}_centre"] = np.array(O.get_centre(), dtype=float)
        scene["optics"][name] = d
    # the render of one placed chain: chain 4 of the C3 scene (the chain of c3_twisted_chain04.npz)
    SourceProperties = {"Divergence": 50e-3 / 2, "SourceSize": 0, "Wavelength": 50e-6, "DeltaFT": 0.5, "NumberRays": 1000}
    Mask = mmask.Mask(msupp.SupportRoundHole(30, 41e-3 / 2 * 500, 0, 0))
    Rr = mmirror.ReturnOptimalToroidalRadii(600, 80)
    Tor = mmirror.MirrorToroidal(Rr[0], Rr[1], msupp.SupportRectangle(200, 30))
    chain = mp.OEPlacement(SourceProperties, [Mask, Tor, Tor], [500, 100, 600], [0, 80, -80],
                           [0, 0, np.linspace(-90, 90, 10)], "2 toroidal mirrors, twisted")[4]
    history = [chain.source_rays] + chain.get_output_rays()
    end = float(np.linalg.norm(chain.source_rays[0].point - chain.optical_elements[0].position))
    for k, seg in enumerate(mplots._RenderRays(history, end, maxRays=5000)):
        arrays[f"c3_segments{k}"] = np.asarray(seg, dtype=float)
    for k, oe in enumerate(chain.optical_elements):
        cloud, _ = mplots._RenderOpticalElement(oe, 250, draw_mesh=False)
        arrays[f"c3_optic{k}"] = np.asarray(cloud, dtype=float)
    scene["c3"] = {"fixture": "c3_twisted_chain04", "EndDistance": end, "OEpoints": 250, "n_history": [len(h) for h in history]}
    arrays["scene_json"] = np.array(json.dumps(scene))
    path = os.path.join(gg.OUT, "render_grids.npz")
    np.savez_compressed(path, **arrays)
    print(f"render_grids: {len(arrays)} arrays, {os.path.getsize(path) / 1024:.0f} kB")


if __name__ == "__main__":
    main()
