"""Alignment-tolerance scan on one resident bundle: pitch the second toroid of the twisted-toroids setup through a
range of angles and watch the focal spot and the pulse duration on a FIXED detector.

The scan re-traces the same optics with new poses many times.  Here that costs, per pose, one small host-to-device
copy and one HIP-graph launch: the element descriptors live in a device-resident scene table (graph.SceneProgram), the
detector read-out rides on the tracing launch, and the only thing that returns to the host is the 24 statistics.
(The reference does the same with `OpticalChain.get_OE_loop_list` + `ARTmain.main`: one full Python trace per pose,
ART/ModuleOpticalChain.py:371-657.)

    python examples/pose_scan.py [rays] [n_poses] [max_pitch_urad]
"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))   # repository root on the path

import ART.ModuleDetector as mdet
import ART.ModuleMask as mmask
import ART.ModuleMirror as mmirror
import ART.ModuleProcessing as mp
import ART.ModuleSupport as msupp
from attosecondraytracing_amd.graph import SceneProgram

rays = int(float(sys.argv[1])) if len(sys.argv) > 1 else 1_000_000
n_poses = int(sys.argv[2]) if len(sys.argv) > 2 else 101
max_urad = float(sys.argv[3]) if len(sys.argv) > 3 else 200.0

source = dict(Divergence=25e-3, SourceSize=0, Wavelength=50e-6, DeltaFT=0.5, NumberRays=rays)
focal, grazing = 600, 80
R, r = mmirror.ReturnOptimalToroidalRadii(focal, grazing)
toroid = mmirror.MirrorToroidal(R, r, msupp.SupportRectangle(200, 30))
mask = mmask.Mask(msupp.SupportRoundHole(30, 10.25, 0, 0))
chain = mp.OEPlacement(source, [mask, toroid, toroid], [500, focal - 500, focal], [0, grazing, -grazing], [0, 0, 0],
                       "mask + 2 toroids")

# the detector of the aligned setup, placed once and then left where it is
aligned = chain.get_output_rays()
detector = mdet.Detector(np.asarray(chain.optical_elements[-1].position, dtype=float))
detector.autoplace(aligned[-1], focal)

program = SceneProgram([chain.source_rays], [chain.optical_elements], detectors=[detector])
light_speed = mdet.LightSpeed
angles = np.linspace(-max_urad, max_urad, n_poses) * 1e-6
spot, duration, alive = [], [], []
t0 = time.perf_counter()
for a in angles:
    scanned = chain.copy_chain()
    scanned.rotate_OE(2, "pitch", np.rad2deg(a))            # the reference's own manipulator (degrees)
    program.update([scanned.optical_elements])
    out = program.run()[0]
    s = detector.readout(out[-1])["stats"]                  # the fused read-out's statistics: 24 doubles to the host
    cnt = s[0]
    var_xy = (s[16] + s[17]) / cnt - (s[6] / cnt) ** 2 - (s[7] / cnt) ** 2
    var_o = s[18] / cnt - (s[1] / cnt) ** 2
    spot.append(np.sqrt(max(var_xy, 0.0)))
    duration.append(np.sqrt(max(var_o, 0.0)) / light_speed * 1e15)
    alive.append(cnt / rays)
dt = time.perf_counter() - t0
print(f"{rays} rays, {n_poses} poses of the second toroid (pitch +-{max_urad:g} urad): {dt * 1e3 / n_poses:.2f} ms per pose "
      f"(host side included)")
for k in range(0, n_poses, max(1, n_poses // 10)):
    print(f"pitch {angles[k] * 1e6:8.1f} urad: spot SD {spot[k] * 1e3:9.3f} um, duration SD {duration[k]:9.4f} fs, "
          f"transmission {100 * alive[k]:5.1f} %")
best = int(np.argmin(spot))
print(f"smallest spot at {angles[best] * 1e6:.1f} urad: {spot[best] * 1e3:.3f} um")
