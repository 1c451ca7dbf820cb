#!/usr/bin/env python3
"""A/B of kernel variants INSIDE ONE PROCESS: the variants (environment knobs the library reads at every launch:
ART_CHAIN_RPL, ART_CHAIN_WAVES) alternate round by round on the same resident data, every launch bracketed by HIP events
on the launch stream, so that clock ramps, thermal state and the box itself are the same for all of them (two bench.py
runs of the same build differ by up to 6 % in kernel time on this pool).

    python tools/ab_kernel.py [--config relay4|C2|C3|C4] [--rounds 12] [--launches 20] [--readout fused|none]
                              [--variants "ART_CHAIN_RPL=1;ART_CHAIN_RPL=2 ART_CHAIN_WAVES=4;LIB=build/variants/libart_x.so"]
(`LIB=path`: a diagnostic BUILD of the library, loaded beside the default one; its results may be wrong by design;
 `RO=none|fused|lite`: the read-out of this variant; lite = the fused tail with 8 instead of 22 statistics,
 ArtChainReadout.lite -- single chains only)
Prints per variant the median / min of the per-round mean launch time and the ratio to the first variant."""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", default="relay4")
    ap.add_argument("--rays", type=int, default=0)
    ap.add_argument("--mirrors", type=int, default=4)
    ap.add_argument("--rounds", type=int, default=12)
    ap.add_argument("--launches", type=int, default=20)
    ap.add_argument("--readout", default="fused", choices=["fused", "none"])
    ap.add_argument("--scene", action="store_true", help="a single chain through the scene-table launch (what a graph-replayed SceneProgram issues) instead of the by-value one")
    ap.add_argument("--mode", default=None, choices=["chain", "element"], help="launch form of a single chain (default: the library's choice)")
    ap.add_argument("--variants", default="ART_CHAIN_RPL=1;ART_CHAIN_RPL=2 ART_CHAIN_WAVES=4")
    args = ap.parse_args()
    import torch
    import bench
    import __graft_entry__
    __graft_entry__.ensure_built()
    from attosecondraytracing_amd import _lib
    import ART.ModuleProcessing as mp
    import ART.ModuleDetector as mdet
    be = _lib.get_backend()
    if args.config == "relay4":
        chain, _ = bench.build_scene(args.mirrors)
        element_lists, kind, dist = [chain.optical_elements], ("point", 0.02), 600.0
        n = args.rays or 10_000_000
    else:
        element_lists, kind, dist = getattr(bench, "scene_" + args.config.lower())()
        n = args.rays or {"C2": 1_000_000, "C3": 10_000_000, "C4": 12_500_000, "C5": 10_000_000}[args.config]
    ign = args.config != "C5"          # C5: Zernike surface with perturbed normals
    src = bench.device_source(n, 0, n, be, kind, 800e-6 if args.config == "C5" else 50e-6)
    dets = []
    for els in element_lists:
        out = mp.RayTracingCalculation(src, els, IgnoreDefects=ign)
        d = mdet.Detector(np.asarray(els[-1].position, dtype=float))
        d.autoplace(out[-1], dist)
        dets.append(d)
        del out
    many = len(element_lists) > 1 or args.scene

    state = {"readout": args.readout}

    def launch():
        fused = state["readout"] == "fused"
        if many:
            return mp.RayTracingCalculationMany([src] * len(element_lists), element_lists, IgnoreDefects=ign,
                                                detectors=dets if fused else None)
        lite = state["readout"] == "lite"
        return mp.RayTracingCalculation(src, element_lists[0], IgnoreDefects=ign, mode=args.mode,
                                        detector=dets[0] if (fused or lite) else None, readout_lite=lite)

    variants = [v.strip() for v in args.variants.split(";")]
    knobs = sorted({kv.split("=")[0] for v in variants for kv in v.split() if "=" in kv} - {"LIB", "RO"})
    times = {v: [] for v in variants}
    default_be, builds = be, {}
    for rnd in range(args.rounds + 1):
        for v in variants:
            for k in knobs:
                os.environ.pop(k, None)
            be = default_be
            state["readout"] = args.readout
            for kv in v.split():
                if kv.startswith("RO="):            # this variant's read-out: fused | none
                    state["readout"] = kv[3:]
                    continue
                if kv.startswith("LIB="):           # another BUILD of the library, loaded into this process beside the default
                    path = kv[4:]
                    if path not in builds:
                        builds[path] = _lib.HipBackend(os.path.join(ROOT, path))
                    be = builds[path]
                elif "=" in kv:
                    k, val = kv.split("=")
                    os.environ[k] = val
            _lib._BACKEND = src._backend = be
            be.trace_events = []
            for _ in range(args.launches):
                launch()
            torch.cuda.synchronize()
            ev, be.trace_events = be.trace_events, None
            if rnd > 0:                                   # round 0 warms every variant up
                times[v].append(float(np.mean([a.elapsed_time(b) for a, b in ev])))
    base = np.median(times[variants[0]])
    print(f"# {args.config} {n} rays, read-out {args.readout}, {args.rounds} rounds x {args.launches} launches per variant, alternating")
    for v in variants:
        t = np.array(times[v])
        print(f"{v or '(default)':48s} median {np.median(t):.4f} ms  min {t.min():.4f}  max {t.max():.4f}  ratio {np.median(t) / base:.3f}")


if __name__ == "__main__":
    main()
