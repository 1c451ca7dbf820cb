#!/usr/bin/env python3
"""In-process A/B of the TRACE phase of the end-to-end C3 workflow (tools/e2e_time.py): `OEPlacement` of the loop list,
then `trace_chain_list(chains, history="lazy")` -- the shared prefix (mask + first toroid, one chain launch) and the suffix
(10 chains x the second toroid, ONE scene launch that also forms the analysis' sums) -- under the library's per-launch
knobs (ART_CHAIN_RPL, ART_CHAIN_WAVES, ART_SCENE_ORDER, ART_SCENE_KEEP), alternating round by round on the same box.
Timed with HIP events around the call (device idle before it: the interval includes the ~0.5 ms of host enqueue that
precedes the last launch, the same for every variant).

    python tools/ab_workflow_trace.py [rays] [--rounds 7] [--variants="NAME=VAL,NAME=VAL;..."] [--nomask]

(`--variants=...` with the equals sign: a list that starts with "-", the default build, would read as an option.)"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import ART.ModuleMask as mmask, ART.ModuleMirror as mmirror, ART.ModuleProcessing as mp, ART.ModuleSupport as msupp
import ART.ModuleOpticalChain as moc

KNOBS = ("ART_CHAIN_RPL", "ART_CHAIN_WAVES", "ART_SCENE_ORDER", "ART_SCENE_KEEP")
DEFAULT = ("-;ART_SCENE_ORDER=chain;ART_SCENE_ORDER=chain,ART_SCENE_KEEP=1;ART_SCENE_KEEP=1;ART_SCENE_KEEP=0;ART_SCENE_ORDER=tile;"
           "ART_CHAIN_RPL=2;ART_CHAIN_RPL=2,ART_SCENE_KEEP=1;ART_CHAIN_WAVES=4")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("rays", nargs="?", type=float, default=1e7)
    ap.add_argument("--rounds", type=int, default=7)
    ap.add_argument("--variants", default=DEFAULT)
    ap.add_argument("--nomask", action="store_true", help="the same list without the mask (prefix toroid 1, suffix 10 x toroid 2): every slot alive")
    args = ap.parse_args()
    source = dict(Divergence=25e-3, SourceSize=0, Wavelength=50e-6, DeltaFT=0.5, NumberRays=int(args.rays))
    R, r = mmirror.ReturnOptimalToroidalRadii(600, 80)
    toroid = mmirror.MirrorToroidal(R, r, msupp.SupportRectangle(200, 30))
    mask = mmask.Mask(msupp.SupportRoundHole(30, 10.25, 0, 0))
    variants = [v.strip() for v in args.variants.split(";")]
    times = {v: [] for v in variants}
    for rnd in range(args.rounds + 1):                       # round 0 warms every variant up
        for v in variants:
            for k in KNOBS:
                os.environ.pop(k, None)
            if v != "-":
                for kv in v.split(","):
                    k, val = kv.split("=")
                    os.environ[k] = val
            if args.nomask:
                chains = mp.OEPlacement(source, [toroid, toroid], [600, 600], [80, -80], [0, np.linspace(-90, 90, 10)], "C3 without mask")
            else:
                chains = mp.OEPlacement(source, [mask, toroid, toroid], [500, 100, 600], [0, 80, -80],
                                        [0, 0, np.linspace(-90, 90, 10)], "C3")
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            moc.trace_chain_list(chains, history="lazy")
            e1.record()
            e1.synchronize()
            if rnd:
                times[v].append(e0.elapsed_time(e1))
            del chains
    for k in KNOBS:
        os.environ.pop(k, None)
    base = np.median(times[variants[0]])
    print(("# WITHOUT the mask: " if args.nomask else "") + f"# C3 workflow, trace phase (prefix launch + suffix scene launch with the sums tail), {int(args.rays)} rays x 10 chains, "
          f"{args.rounds} rounds, in-process A/B")
    for v in variants:
        t = np.array(times[v])
        print(f"{v:42s} median {np.median(t):.4f} ms  min {t.min():.4f}  max {t.max():.4f}  ratio {np.median(t) / base:.3f}", flush=True)


if __name__ == "__main__":
    main()
