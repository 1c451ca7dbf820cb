"""Where the time of the end-to-end C3 workflow (examples/twisted_toroids_large.py) goes: scene construction, the one
batched trace of all chains, and the analysis of the loop list (transmission, detector placement, autofocus), cold and
warm pass (the second pass reuses the caching allocator's blocks).

    python tools/e2e_time.py [rays] [batched|lazy|loop] [--profile] [--passes K]

`--profile`: cProfile of the warm pass's construction and analysis (top of the cumulative list)."""
import cProfile, os, pstats, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import ART.ModuleMask as mmask, ART.ModuleMirror as mmirror, ART.ModuleProcessing as mp, ART.ModuleSupport as msupp
import ART.ModuleOpticalChain as moc
import ARTmain

args = [a for a in sys.argv[1:] if not a.startswith("--")]
rays = int(float(args[0])) if len(args) > 0 else 10_000_000
how = args[1] if len(args) > 1 else "lazy"       # batched | lazy (batched, only the analysed bundle written) | loop
profile = "--profile" in sys.argv
passes = int(sys.argv[sys.argv.index("--passes") + 1]) if "--passes" in sys.argv else 3
batched = how in ("batched", "lazy")
kw = {"history": "lazy"} if how == "lazy" else {}
source = dict(Divergence=25e-3, SourceSize=0, Wavelength=50e-6, DeltaFT=0.5, NumberRays=rays)
R, r = mmirror.ReturnOptimalToroidalRadii(600, 80)
toroid = mmirror.MirrorToroidal(R, r, msupp.SupportRectangle(200, 30))
mask = mmask.Mask(msupp.SupportRoundHole(30, 10.25, 0, 0))
SP, DO, AO = ARTmain.complete_defaults(source, dict(ReflectionNumber=-1, ManualDetector=False, DistanceDetector=600,
                                                    AutoDetectorDistance=True, OptFor="intensity"),
                                       dict(verbose=False, save_results=False))


def summary(res):
    return np.array([[d.get_distance(), t, s, u] for (_, d, t, s, u) in res])


def analyse(chains):
    """The analysis as ARTmain.main runs it: all chains of the list at once where the package offers that."""
    if hasattr(ARTmain, "analyse_chain_list") and batched:
        return ARTmain.analyse_chain_list(chains, SP, DO, AO)
    return [ARTmain.run_ART(ch, SP, DO, AO, True) for ch in chains]


import gc
for rep in range(passes):
    if rep == 1:
        # Python's cyclic collector: a full pass walks the ~1e6 objects torch / numpy leave behind and stops the host for
        # 40-60 ms at an arbitrary point (a warm pass once read "trace 63.5 ms"): what is alive after the cold pass moves
        # to the permanent generation, as bench.py does before its timed region
        gc.collect()
        gc.freeze()
    pr = cProfile.Profile() if (profile and rep == passes - 1) else None
    t0 = time.perf_counter()
    if pr: pr.enable()
    chains = mp.OEPlacement(source, [mask, toroid, toroid], [500, 100, 600], [0, 80, -80], [0, 0, np.linspace(-90, 90, 10)], "C3")
    if pr: pr.disable()
    torch.cuda.synchronize(); t1 = time.perf_counter()
    if batched:
        moc.trace_chain_list(chains, **kw)
    else:
        for ch in chains:
            ch.get_output_rays()
    torch.cuda.synchronize(); t2 = time.perf_counter()
    if pr: pr.enable()
    res = analyse(chains)
    if pr: pr.disable()
    torch.cuda.synchronize(); t3 = time.perf_counter()
    print(f"pass {rep} ({'one scene launch' + (', lazy history' if kw else '') if batched else 'chain by chain'}): construction {1e3*(t1-t0):.1f} ms, trace of 10 chains "
          f"{1e3*(t2-t1):.1f} ms, analysis {1e3*(t3-t2):.1f} ms ({1e2*(t3-t2):.2f} ms per chain)", flush=True)
    if rep == passes - 1:
        got = summary(res)
        # the same chains analysed one by one (ARTmain.run_ART): the list analysis must give the same numbers
        one = summary([ARTmain.run_ART(ch, SP, DO, AO, True) for ch in chains])
        worst = np.abs(got - one).max(axis=0)
        print("list analysis vs chain-by-chain run_ART: max |diff| distance %.3g mm, transmission %.3g %%, spot %.3g mm, "
              "duration %.3g fs" % tuple(worst))
        assert np.array_equal(got, one), "the list analysis differs from run_ART chain by chain"
        for row in got:
            print("   distance %.3f mm  transmission %.2f %%  spot sd %.4g um  duration sd %.4g fs" % (row[0], row[1], 1e3 * row[2], row[3]))
    if pr:
        pstats.Stats(pr).sort_stats("cumulative").print_stats(45)
        pstats.Stats(pr).sort_stats("tottime").print_stats(25)
    del chains, res

# The same workflow as a program runs it: NO synchronisation between the phases (construction, trace and analysis queue up
# behind each other on the device while the host goes on; the one wait is the analysis' copy of its 64 doubles per chain).
# The phase times above each end in a torch.cuda.synchronize(): their sum counts device work the host does not wait for.
whole, marks = [], []
for rep in range(max(passes, 5)):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    chains = mp.OEPlacement(source, [mask, toroid, toroid], [500, 100, 600], [0, 80, -80], [0, 0, np.linspace(-90, 90, 10)], "C3")
    t1 = time.perf_counter()
    if batched:
        moc.trace_chain_list(chains, **kw)
    else:
        for ch in chains:
            ch.get_output_rays()
    t2 = time.perf_counter()
    res = analyse(chains)
    t3 = time.perf_counter()
    torch.cuda.synchronize()
    whole.append(1e3 * (time.perf_counter() - t0))
    marks.append((1e3 * (t1 - t0), 1e3 * (t2 - t1), 1e3 * (t3 - t2)))
    del chains, res
print("whole workflow, one synchronisation at the end: " + " ".join(f"{t:.2f}" for t in whole) + f" ms; median {np.median(whole):.2f} ms", flush=True)
m = np.median(np.array(marks), axis=0)
print(f"   host time until each phase RETURNS (no synchronisation; the analysis waits for its result): construction {m[0]:.2f} ms, "
      f"trace enqueue {m[1]:.2f} ms, analysis {m[2]:.2f} ms", flush=True)
