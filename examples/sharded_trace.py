"""Rays sharded over the GPUs of one node (one process per GPU, RCCL): every rank generates and traces its own index
range of one big point source; the ranks exchange their read-out statistics and a sample of the read-out in one
all-gather (sharding.Exchange), then rank 0 gathers the surviving rays' (number, X, Y, optical path) records -- 28 bytes
per survivor, one collective (sharding.SurvivorGather) -- and prints the global result.

    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 --master-port 29511 \\
        examples/sharded_trace.py [total_rays]
(also runs with --nproc-per-node 1)
"""
import datetime
import os
import sys

# Multi-process GPU jobs on this pool need the ROCr runtime's dmabuf IPC: RCCL opens its peers' buffers through
# hipIpcGetMemHandle / hipIpcOpenMemHandle, which fail with "invalid argument" in the runtime's LEGACY IPC mode as soon as
# two ranks of one node set up their xGMI transport (bench.py: multi_process_env).  Set before torch initialises HIP;
# setdefault: an explicit choice in the caller's environment wins.
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

import numpy as np
import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))   # repository root on the path

rank, world, local = (int(os.environ.get(k, d)) for k, d in (("RANK", 0), ("WORLD_SIZE", 1), ("LOCAL_RANK", 0)))
torch.cuda.set_device(local)
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29511")
dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local),
                        timeout=datetime.timedelta(seconds=300))      # a rank that never arrives ends the job, not hangs it

import ART.ModuleMask as mmask                      # noqa: E402
import ART.ModuleMirror as mmirror                  # noqa: E402
import ART.ModuleProcessing as mp                   # noqa: E402
import ART.ModuleSupport as msupp                   # noqa: E402
import ART.ModuleDetector as mdet                   # noqa: E402
from attosecondraytracing_amd import _lib, sharding, ModuleGeometry as mgeo   # noqa: E402
from attosecondraytracing_amd.bundle import RayBundle                         # noqa: E402

n_total = int(float(sys.argv[1])) if len(sys.argv) > 1 else 40_000_000
be = _lib.get_backend()

# the scene of CONFIG_2toroidals_twisted (one twist angle), placed with a small bundle: identical on every rank
focal, grazing = 600, 80
R, r = mmirror.ReturnOptimalToroidalRadii(focal, grazing)
toroid = mmirror.MirrorToroidal(R, r, msupp.SupportRectangle(200, 30))
mask = mmask.Mask(msupp.SupportRoundHole(30, 10.25, 0, 0))
small = dict(Divergence=25e-3, SourceSize=0, Wavelength=50e-6, DeltaFT=0.5, NumberRays=1000)
chain = mp.OEPlacement(small, [mask, toroid, toroid], [500, focal - 500, focal], [0, grazing, -grazing], [0, 0, 30.0])

# this rank's shard of the big source: global ray indices [lo, hi), generated on the device
lo, hi = sharding.shard_range(n_total, rank, world)
src = RayBundle.allocate(hi - lo, backend=be)
src.wavelength = 50e-6
rot = mgeo.rotation_matrix(np.array([0.0, 0.0, 1.0]), np.array([1.0, 0.0, 0.0]))
be.make_source(0, 25e-3, rot, np.zeros(3), lo, hi - lo, n_total, src.view())
src.intensity = torch.ones(hi - lo, dtype=torch.float64, device=be.device)

out = mp.RayTracingCalculation(src, chain.optical_elements)
# the detector must be the same on every rank: place it from the small aligned bundle of the chain
det = mdet.Detector(np.asarray(chain.optical_elements[-1].position, dtype=float))
det.autoplace(chain.get_output_rays()[-1], focal)
ro = det.readout(out[-1], sync=False)
exchange = sharding.Exchange(be, hi - lo, sample=20000)
stats, sample = exchange(ro["stats_dev"], ro["X"], ro["Y"], ro["opl"], out[-1].alive)
# every SURVIVING ray's read-out to rank 0, in global ray order
specs = [sharding.shard_spec(n_total, rk, world) for rk in range(world)]
gather = sharding.SurvivorGather(be, hi - lo, world, rank, dst=0, buffers=1, specs=specs)
sent = gather.start(0, ro["X"], ro["Y"], ro["opl"], out[-1].alive)
gather.drain()
s = stats.cpu().numpy()
if rank == 0:
    number, X, Y, path = gather.assemble(0)
    assert number.numel() == int(s[0]) and bool((number[1:] > number[:-1]).all())
    delays_fs = (path - path.mean()) / mdet.LightSpeed * 1e15           # Detector.get_Delays, ART/ModuleDetector.py:272-279
    print(f"gathered {number.numel()} survivor records ({sent} B per rank): delay std {float(delays_fs.std()):.4f} fs")
    count, mean_path = s[0], s[1] / s[0]
    var_x = s[16] / count - (s[6] / count) ** 2
    var_y = s[17] / count - (s[7] / count) ** 2
    print(f"{world} GPU(s), {n_total} rays: {int(count)} reach the detector ({100 * count / n_total:.2f} %), mean optical "
          f"path {mean_path:.6f} mm, spot std {np.sqrt(var_x + var_y) * 1e3:.3f} um, sample for plots: {tuple(sample.shape)}")
dist.barrier()
dist.destroy_process_group()
