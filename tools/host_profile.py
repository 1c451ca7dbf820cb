#!/usr/bin/env python3
"""Where the HOST time of the C3 workflow goes (OEPlacement of the loop list -> trace_chain_list -> analyse_chain_list):
cProfile over many passes at a small ray count (the device work is then negligible; the Python work per pass does not
depend on the ray count), printed as microseconds PER PASS.

    python tools/host_profile.py [rays] [passes] [top]"""
import cProfile
import os
import pstats
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import ART.ModuleMask as mmask, ART.ModuleMirror as mmirror, ART.ModuleProcessing as mp, ART.ModuleSupport as msupp
import ART.ModuleOpticalChain as moc
import ARTmain

rays = int(float(sys.argv[1])) if len(sys.argv) > 1 else 100_000
passes = int(sys.argv[2]) if len(sys.argv) > 2 else 100
top = int(sys.argv[3]) if len(sys.argv) > 3 else 30
source = dict(Divergence=25e-3, SourceSize=0, Wavelength=50e-6, DeltaFT=0.5, NumberRays=rays)
R, r = mmirror.ReturnOptimalToroidalRadii(600, 80)
toroid = mmirror.MirrorToroidal(R, r, msupp.SupportRectangle(200, 30))
mask = mmask.Mask(msupp.SupportRoundHole(30, 10.25, 0, 0))
SP, DO, AO = ARTmain.complete_defaults(source, dict(ReflectionNumber=-1, ManualDetector=False, DistanceDetector=600,
                                                    AutoDetectorDistance=True, OptFor="intensity"),
                                       dict(verbose=False, save_results=False))


def once():
    t0 = time.perf_counter()
    chains = mp.OEPlacement(source, [mask, toroid, toroid], [500, 100, 600], [0, 80, -80], [0, 0, np.linspace(-90, 90, 10)], "C3")
    t1 = time.perf_counter()
    moc.trace_chain_list(chains, history="lazy")
    t2 = time.perf_counter()
    ARTmain.analyse_chain_list(chains, SP, DO, AO)
    t3 = time.perf_counter()
    return t1 - t0, t2 - t1, t3 - t2


import gc
for _ in range(5):
    once()
gc.collect()
gc.freeze()
ts = np.array([once() for _ in range(passes)])
print("un-profiled, median over %d passes at %d rays: construction %.0f us, trace enqueue %.0f us, analysis (with its wait) %.0f us"
      % ((passes, rays) + tuple(1e6 * np.median(ts, axis=0))), flush=True)
prs = [cProfile.Profile() for _ in range(3)]          # one profile per phase
for _ in range(passes):
    prs[0].enable()
    chains = mp.OEPlacement(source, [mask, toroid, toroid], [500, 100, 600], [0, 80, -80], [0, 0, np.linspace(-90, 90, 10)], "C3")
    prs[0].disable()
    prs[1].enable()
    moc.trace_chain_list(chains, history="lazy")
    prs[1].disable()
    prs[2].enable()
    ARTmain.analyse_chain_list(chains, SP, DO, AO)
    prs[2].disable()
    del chains
for phase, pr in zip(("construction (OEPlacement)", "trace enqueue (trace_chain_list)", "analysis (analyse_chain_list)"), prs):
    st = pstats.Stats(pr)
    rows = sorted(st.stats.items(), key=lambda kv: -kv[1][2])[:top]        # by tottime
    total = sum(v[2] for v in st.stats.values())
    print("\n== %s: %.0f us per pass under the profiler" % (phase, 1e6 * total / passes))
    print("%9s %9s %8s  function" % ("tot us", "cum us", "calls"))
    for (fn, line, name), (cc, nc, tt, ct, _) in rows:
        print("%9.1f %9.1f %8.1f  %s:%d(%s)" % (1e6 * tt / passes, 1e6 * ct / passes, nc / passes, os.path.basename(fn), line, name))
