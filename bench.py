#!/usr/bin/env python3
"""bench.py -- ray-surface intersections/s on MI355X (BASELINE.json metric), one JSON line on stdout.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--config relay4|C2|C3|C4|C5]

Default workload ("relay4", the headline BASELINE.json's metric is quoted on): a point source (half-angle 20 mrad) of
1e7 rays per GPU through 4 toroidal mirrors (two f-x-f relays with the C3 scene's toroid: f = 600 mm, 80 deg, 200 x 30 mm),
then the detector read-out: 4e7 ray-surface intersections per GPU and step, every ray surviving, full per-element
history written as the API returns it; the step is replayed from a HIP graph (graph.SceneProgram; `--graph off`: eager
launches).  `--config` selects the other BASELINE.json configurations (same JSON contract):
  C2  CONFIG_2toroidals_f-x-f: 11 chains (loop list over the toroid distance) x 1e6 rays x (mask + 2 toroids), traced by
      ONE launch from a device-resident scene table, + 11 read-outs
  C3  CONFIG_2toroidals_twisted: 10 chains (incidence-plane twist) x 1e7 rays x (mask + 2 toroids) + read-outs, one launch
  C4  8-element mixed chain (OAP, plane, 2 toroids, 2 planes, OAP, plane), 1.25e7 rays per GPU (1e8 over 8 GPUs)
  C5  CONFIG_deformed geometry with a 6th-order Zernike defect, IgnoreDefects=False (perturbed normals), 1e7 rays

A step = one pass of the hot path over resident bundles:
    RayTracingCalculation(source, elements)   every per-element bundle written
    Detector.readout(last)                    X, Y, optical path per ray + 24 global statistics (fused reductions)
    N > 1:  + the gather BASELINE.json's north_star names, in EVERY step: (number:int32, X, Y, optical path) of every
            SURVIVING ray of the analysed chain to rank 0 (28 B per survivor; 24 B from a shard that loses
            nothing -- its read-out writes straight into the send buffer: zero-copy), as point-to-point transfers over
            RCCL (one per peer into the root), behind ONE 208-byte all-gather of every shard's count + 24 statistics
            (the delays are relative to the GLOBAL mean path, ART/ModuleDetector.py:277), double-buffered behind the
            next step's tracing -- this is `value` (round 5; `value_full_gather` repeats it for older readers);
            and, in a second timed region of the same K steps, the step with ONE all-gather of the statistics and an
            evenly spaced 20000-ray sample instead (what a plot draws) -- `value_stats_exchange`.
Beside `value` (every per-element bundle written) the line carries `value_lazy_history`: the same step with only the
analysed bundle written, the product's lazy-history mode (what ARTmain uses) -- never the headline.  `roofline.frac` is
the TIMED REGION's: counted HBM bytes (committed rocprofv3 PMC profile of THIS build, matched by source hash; else the
compulsory bytes computed in the run: `frac_basis`) x launches per step / ms_per_step / 8 TB/s; `frac_post_region` divides
by event-bracketed launches issued after the region.  `cpu_baseline` is the oracle timed on a bounded sample on this
host (`reference_as_is`: the reference's own loops, timed in the build container); `box` holds the clocks / power /
partition mode of the device before the run and under load.

Inputs are resident in HBM before the timed region; nothing is copied to the host inside a step.  N > 1 is weak scaling:
every rank traces its own shard (index range of an N x rays source), no collective on the tracing path.

`--gpus N` with N > 1 and no WORLD_SIZE in the environment: this process is only a LAUNCHER -- it starts N worker
processes of itself (one per GPU, RANK/LOCAL_RANK/WORLD_SIZE/MASTER_* set, free rendezvous port on 127.0.0.1) before
anything touches the GPU, relays rank 0's JSON line and exits non-zero if a worker fails.  Under torchrun (WORLD_SIZE set)
it is a worker.  A worker fails if the process group's size differs from --gpus.

This file is the entry point; its parts live in tools/bench/ (launcher, workloads, baselines, roofline, worker).
"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

# the parts (re-exported: tools/*.py and the tests use bench.build_scene, bench.scene_c3, bench.device_source, ...)
from tools.bench.launcher import launch_workers, multi_process_env, BoxState, log  # noqa: E402,F401
from tools.bench.workloads import CONFIGS, build_scene, scene_c2, scene_c3, scene_c4, scene_c5, device_source  # noqa: E402,F401
from tools.bench.roofline import profiled_traffic, HBM_PEAK_GBS, XGMI_LINK_GBS  # noqa: E402,F401
from tools.bench.baselines import cpu_baseline, parity_against, cpu_twin_allcores, oracle_elements  # noqa: E402,F401


def main(argv=None):
    argv = sys.argv[1:] if argv is None else argv
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--config", default="relay4", choices=CONFIGS, help="BASELINE.json configuration (default: the headline)")
    ap.add_argument("--rays", type=int, default=0, help="rays per GPU (0 = the configuration's own size)")
    ap.add_argument("--mirrors", type=int, default=4, help="relay4 only: number of toroidal mirrors")
    ap.add_argument("--mode", default=None, choices=[None, "chain", "element"])
    ap.add_argument("--graph", default="auto", choices=["auto", "on", "off"],
                    help="replay the step from a HIP graph (auto = on); off: eager launches")
    ap.add_argument("--shard", default="blocks", choices=["blocks", "strided"],
                    help="N > 1: contiguous index ranges per rank (default) or rank r traces rays r, r + N, ...")
    ap.add_argument("--readout", default="auto", choices=["auto", "fused", "separate", "lite"],
                    help="fused: the detector read-out rides on the tracing launch; separate: its own kernel afterwards; "
                         "auto (default) = fused; lite: fused, but only 8 of the 22 statistics are reduced (what a plain "
                         "get_Delays / get_PointList2DCentre caller consumes) -- a measurement option, never the default")
    ap.add_argument("--gather-copy", default="zero", choices=["zero", "pack"],
                    help="N > 1: zero (default) = a shard that loses nothing writes its read-out straight into the gather's "
                         "send buffer; pack = always compact through art_pack_survivors (the round-4 path, for A/B)")
    ap.add_argument("--gather-tiles", type=int, default=1,
                    help="N > 1, zero-copy shards: trace the step as T launches over consecutive slot ranges and send each "
                         "range's records while the next is traced (default 1: the whole step's records leave behind it)")
    ap.add_argument("--cpu-sample", type=int, default=-1, help="rays of the CPU-baseline sample (0 = skip)")
    ap.add_argument("--placement-tries", type=int, default=1,
                    help="opt-in: let the step's program time its launch into N candidate output allocations and keep the "
                         "fastest (graph.SceneProgram; default 1 = take the first allocation)")
    ap.add_argument("--preheat-ms", type=float, default=0.0,
                    help="opt-in (default 0 = the contract's region as it is: W warm-up steps after an idle device): keep the "
                         "device busy with untimed steps for this long before the W warm-up steps, so that the timed region "
                         "runs at sustained clocks; reported as `preheat_ms`")
    ap.add_argument("--time-limit", type=float, default=float(os.environ.get("ART_BENCH_TIME_LIMIT", "1500")),
                    help="N > 1 launcher: wall-clock limit in seconds after which all workers are killed (exit code 4)")
    ap.add_argument("--pg-timeout", type=float, default=float(os.environ.get("ART_PG_TIMEOUT", "300")),
                    help="N > 1 worker: timeout in seconds of the process group's collectives")
    args = ap.parse_args(argv)
    if args.cpu_sample < 0:
        # a bounded sample of the same workload: ~5 s of the single-threaded oracle, so that the GPU part is a visible share
        # of the driver's run (round 4: 2e6 rays = 20 s of a 24-s run)
        args.cpu_sample = {"relay4": 500_000, "C2": 500_000, "C3": 500_000, "C4": 200_000, "C5": 150_000}[args.config]
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        return launch_workers(args.gpus, argv, args.time_limit)      # nothing above or in there touches the GPU
    from tools.bench.worker import worker
    return worker(args)


if __name__ == "__main__":
    sys.exit(main())
