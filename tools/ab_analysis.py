#!/usr/bin/env python3
"""In-process A/B of the list analysis' moments pass (art_analyse_bundles with J jobs that bring their sums along: place +
ONE moments pass + fold): XCD-grouped workgroup order (the default) against job-major (ART_ANALYSIS_ORDER=job, read per
call), alternating round by round on the same resident bundles -- J different bundles sharing one intensity array, like
the chains of a loop list.

    python tools/ab_analysis.py [rays] [--jobs 10] [--tail 0.145] [--rounds 9]"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("rays", nargs="?", type=float, default=1e7)
    ap.add_argument("--jobs", type=int, default=10)
    ap.add_argument("--tail", type=float, default=0.145)
    ap.add_argument("--rounds", type=int, default=9)
    args = ap.parse_args()
    import bench
    from attosecondraytracing_amd import _lib, _abi, analysis
    from attosecondraytracing_amd.bundle import RayBundle
    be = _lib.get_backend()
    n = int(args.rays)
    src = bench.device_source(n, 0, n, be, ("point", 0.02))
    src.intensity = torch.rand(n, dtype=torch.float64, device=be.device) + 0.5
    b = RayBundle.allocate(n, like=src, backend=be)
    b.data.copy_(src.data)
    b.data[0:3] += 600.0 * src.data[3:6]
    b.data[6] = 600.0
    b.alive.fill_(1)
    if args.tail > 0:
        b.alive[int(n * (1.0 - args.tail)):] = 0
    bundles = [b] + [b.copy() for _ in range(args.jobs - 1)]
    jobs = []
    for o in bundles:
        o.intensity = src.intensity
        j = analysis._job(o, _abi.ART_JOB_AUTOPLACE, 100.0)
        j.sums = be.bundle_sums9(o).data_ptr()          # (the sums a tracing launch's tail would have formed)
        jobs.append(j)
    keep = [be.bundle_sums9(o) for o in bundles]
    for j, k in zip(jobs, keep):
        j.sums = k.data_ptr()
    variants = ["xcd", "job"]
    times = {v: [] for v in variants}
    for rnd in range(args.rounds + 1):
        for v in variants:
            os.environ["ART_ANALYSIS_ORDER"] = v
            for _ in range(2):
                be.analyse_bundles(jobs, n)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(5):
                out = be.analyse_bundles(jobs, n)
            e1.record()
            e1.synchronize()
            if rnd:
                times[v].append(e0.elapsed_time(e1) / 5)
    os.environ.pop("ART_ANALYSIS_ORDER", None)
    base = np.median(times["job"])
    print(f"# art_analyse_bundles, {args.jobs} jobs x {n} rays (last {args.tail:.3f} dead), sums given: place + moments + fold, in-process A/B")
    for v in variants:
        t = np.array(times[v])
        print(f"order {v:4s}: median {np.median(t):.4f} ms  min {t.min():.4f}  max {t.max():.4f}  ratio {np.median(t) / base:.3f}", flush=True)


if __name__ == "__main__":
    main()
