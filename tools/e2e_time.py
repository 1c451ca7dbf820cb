"""Where the time of the end-to-end C3 workflow (examples/twisted_toroids_large.py) goes: scene construction, the one
batched trace of all chains, and the per-chain analysis (transmission, detector placement, autofocus), first and second
pass (the second pass reuses the caching allocator's blocks)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import ART.ModuleMask as mmask, ART.ModuleMirror as mmirror, ART.ModuleProcessing as mp, ART.ModuleSupport as msupp
import ART.ModuleOpticalChain as moc
import ARTmain

rays = int(float(sys.argv[1])) if len(sys.argv) > 1 else 10_000_000
how = sys.argv[2] if len(sys.argv) > 2 else "batched"       # batched | lazy (batched, only the analysed bundle written) | loop
batched = how in ("batched", "lazy")
kw = {"history": "lazy"} if how == "lazy" else {}
source = dict(Divergence=25e-3, SourceSize=0, Wavelength=50e-6, DeltaFT=0.5, NumberRays=rays)
R, r = mmirror.ReturnOptimalToroidalRadii(600, 80)
toroid = mmirror.MirrorToroidal(R, r, msupp.SupportRectangle(200, 30))
mask = mmask.Mask(msupp.SupportRoundHole(30, 10.25, 0, 0))
SP, DO, AO = ARTmain.complete_defaults(source, dict(ReflectionNumber=-1, ManualDetector=False, DistanceDetector=600,
                                                    AutoDetectorDistance=True, OptFor="intensity"),
                                       dict(verbose=False, save_results=False))
for rep in range(2):
    t0 = time.perf_counter()
    chains = mp.OEPlacement(source, [mask, toroid, toroid], [500, 100, 600], [0, 80, -80], [0, 0, np.linspace(-90, 90, 10)], "C3")
    torch.cuda.synchronize(); t1 = time.perf_counter()
    if batched:
        moc.trace_chain_list(chains, **kw)
    else:
        for ch in chains:
            ch.get_output_rays()
    torch.cuda.synchronize(); t2 = time.perf_counter()
    res = [ARTmain.run_ART(ch, SP, DO, AO, True) for ch in chains]
    torch.cuda.synchronize(); t3 = time.perf_counter()
    print(f"pass {rep} ({'one scene launch' + (', lazy history' if kw else '') if batched else 'chain by chain'}): construction {1e3*(t1-t0):.1f} ms, trace of 10 chains "
          f"{1e3*(t2-t1):.1f} ms, analysis {1e3*(t3-t2):.1f} ms ({1e2*(t3-t2):.2f} ms per chain)", flush=True)
    del chains, res
