"""Two toroidal mirrors behind an annular mask, the second one twisted about the beam axis, traced with a large
bundle and analysed like ARTmain does (energy transmission, detector autoplacement, autofocus).  The optical setup is
the one of the reference's CONFIG_2toroidals_twisted example; everything here goes through the public API.

    python examples/twisted_toroids_large.py [rays] [n_twist_angles]
"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))   # repository root on the path

import ART.ModuleMask as mmask
import ART.ModuleMirror as mmirror
import ART.ModuleProcessing as mp
import ART.ModuleSupport as msupp
from ARTmain import main

rays = int(float(sys.argv[1])) if len(sys.argv) > 1 else 10_000_000
n_angles = int(sys.argv[2]) if len(sys.argv) > 2 else 10

source = dict(Divergence=25e-3, SourceSize=0, Wavelength=50e-6, DeltaFT=0.5, NumberRays=rays)
focal, grazing = 600, 80
R, r = mmirror.ReturnOptimalToroidalRadii(focal, grazing)
toroid = mmirror.MirrorToroidal(R, r, msupp.SupportRectangle(200, 30))
mask = mmask.Mask(msupp.SupportRoundHole(30, 10.25, 0, 0))

t0 = time.perf_counter()
chains = mp.OEPlacement(source, [mask, toroid, toroid], [500, focal - 500, focal], [0, grazing, -grazing],
                        [0, 0, np.linspace(-90, 90, n_angles)], "mask + 2 toroids, second incidence plane twisted")
t1 = time.perf_counter()
detector = dict(ReflectionNumber=-1, ManualDetector=False, DistanceDetector=focal, AutoDetectorDistance=True,
                OptFor="intensity")
analysis = dict(verbose=False, save_results=False)
kept = main(chains, source, detector, analysis)
t2 = time.perf_counter()
print(f"\n{rays} rays x {n_angles} chains: scene construction {t1 - t0:.2f} s, trace + analysis {t2 - t1:.2f} s "
      f"({(t2 - t1) / n_angles * 1e3:.1f} ms per chain)")
for ch, det, et, s, d in zip(kept["OpticalChain"], kept["Detector"], kept["ETransmission"], kept["SpotSizeSD"],
                             kept["DurationSD"]):
    print(f"twist {ch.loop_variable_value:7.2f} deg: transmission {et:5.2f} %, focus at {det.get_distance():8.3f} mm, "
          f"spot {s * 1e3:8.3f} um, duration {d:8.4f} fs")
