#!/usr/bin/env python3
"""Golden vectors of the ANALYSIS behind the trace, produced by RUNNING THE REFERENCE here:
for EVERY chain of the two shipped loop lists -- C2 (examples/CONFIG_2toroidals_f-x-f.py: 11 chains) and C3
(examples/CONFIG_2toroidals_twisted.py: 10 chains) -- what the reference's launcher computes per chain
(ARTmain.run_ART, ART/ARTmain.py:248-300):
    mplots.getETransmission                    ART/ModuleAnalysisAndPlots.py:62-77
    Detector.autoplace(DistanceDetector)       ART/ModuleDetector.py:109-137
    mplots.GetResultSummary                    ART/ModuleAnalysisAndPlots.py:81-129   (AutoDetectorDistance = False)
    mp.FindOptimalDistance(..., "intensity", None, 3, IntensityWeighted=True)          (AutoDetectorDistance = True:
                                               ART/ARTmain.py:147-190 -> ART/ModuleProcessing.py:369-460)
plus two further FindOptimalDistance variants ("intensity" unweighted, "duration" weighted).  1000 source rays per
chain: every bundle has <= 1000 rays, so optimize_detector's random 1000-ray subsample (ARTmain.py:168-171) never
happens -- the search runs on the FULL ray set, nothing is seeded.

TEST INFRASTRUCTURE, same rules as generate_goldens.py (whose stand-ins and object descriptions it reuses): run once
in the build container (`python tests/golden/generate_analysis_goldens.py`), commit the two small .npz files; nothing on
the GPU box imports this script or the reference."""
import json
import os
import sys
import time

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import generate_goldens as gg  # noqa: E402  (sets up sys.path for the reference + stand-ins, imports ART.*)

import numpy as np  # noqa: E402

mp, mdet, mplots, mmirror, mmask, msupp = gg.mp, gg.mdet, gg.mplots, gg.mmirror, gg.mmask, gg.msupp


def analyse_list(name, chains, distance):
    arrays = {}
    gg.bundle_arrays(chains[0].source_rays, "src_", arrays, with_path=False)
    wl = chains[0].source_rays[0].wavelength
    scene = {"name": name, "wavelength": float(wl), "detector_distance": float(distance), "chains": []}
    for i, ch in enumerate(chains):
        t0 = time.perf_counter()
        last = ch.get_output_rays()[-1]
        e = {"loop_variable_value": float(ch.loop_variable_value), "elements": []}
        for k, oe in enumerate(ch.optical_elements):
            d = gg.describe_optic(oe.type, arrays, f"c{i}_el{k}_")
            d["position"] = [float(v) for v in oe.position]
            d["normal"] = [float(v) for v in oe.normal]
            d["majoraxis"] = [float(v) for v in oe.majoraxis]
            e["elements"].append(d)
        arrays[f"c{i}_last_number"] = np.array([r.number for r in last], dtype=np.int64)
        e["ETransmission"] = float(mplots.getETransmission(ch.source_rays, last))
        det = mdet.Detector(ch.optical_elements[-1].position)
        det.autoplace(last, distance)
        e["detector"] = {"centre": [float(v) for v in det.centre], "normal": [float(v) for v in det.normal],
                         "refpoint": [float(v) for v in det.refpoint], "distance": float(det.get_distance())}
        spot, dur = mplots.GetResultSummary(det, last, False)
        e["SpotSizeSD"], e["DurationSD"] = float(spot), float(dur)
        e["NA"] = float(mp.ReturnNumericalAperture(last, 1))
        e["autofocus"] = {}
        for optfor, weighted in (("intensity", True), ("intensity", False), ("duration", True)):
            D, s, t = mp.FindOptimalDistance(det, last, optfor, None, 3, weighted, False)
            e["autofocus"][f"{optfor}_{int(weighted)}"] = [float(D.get_distance()), float(s), float(t)]
        scene["chains"].append(e)
        print(f"{name} chain {i}: {len(last)} rays, ET {e['ETransmission']:.3f} %, spot {spot:.6g} mm, dur {dur:.6g} fs, "
              f"optimum {e['autofocus']['intensity_1']}  ({time.perf_counter() - t0:.1f} s)", flush=True)
    arrays["scene_json"] = np.array(json.dumps(scene))
    path = os.path.join(gg.OUT, name + ".npz")
    np.savez_compressed(path, **arrays)
    print(f"{name}: {len(chains)} chains, {os.path.getsize(path) / 1024:.0f} kB", flush=True)


def main():
    gg.check_standin_rotations()
    n_rays = 1000
    SP = {"Divergence": 50e-3 / 2, "SourceSize": 0, "Wavelength": 50e-6, "DeltaFT": 0.5, "NumberRays": n_rays}
    # C2: examples/CONFIG_2toroidals_f-x-f.py:19-68
    Mask = mmask.Mask(msupp.SupportRoundHole(20, 14e-3 * 500, 0, 0))
    R, r = mmirror.ReturnOptimalToroidalRadii(500, 80)
    Tor = mmirror.MirrorToroidal(R, r, msupp.SupportRectangle(150, 32))
    chains = mp.OEPlacement(dict(SP), [Mask, Tor, Tor], [400, 100, np.linspace(300, 700, 11)], [0, 80, -80], [0, 0, 0],
                            "2 toroidal mirrors in f-d-f config")
    analyse_list("analysis_c2", chains, 500)
    # C3: examples/CONFIG_2toroidals_twisted.py:19-67
    Mask = mmask.Mask(msupp.SupportRoundHole(30, 41e-3 / 2 * 500, 0, 0))
    R, r = mmirror.ReturnOptimalToroidalRadii(600, 80)
    Tor = mmirror.MirrorToroidal(R, r, msupp.SupportRectangle(200, 30))
    chains = mp.OEPlacement(dict(SP), [Mask, Tor, Tor], [500, 100, 600], [0, 80, -80], [0, 0, np.linspace(-90, 90, 10)],
                            "2 toroidal mirrors, twisted")
    analyse_list("analysis_c3", chains, 600)


if __name__ == "__main__":
    main()
