#!/usr/bin/env python3
"""Time the reduction / compaction entry points of libart_hip.so on one resident bundle (1e7 rays by default), each call
bracketed by HIP events over `reps` back-to-back calls, and print microseconds per call and the HBM rate its COMPULSORY
bytes give (the bytes the call must move: what it reads of the bundle + what it writes; partial-statistics traffic is
noise).  The per-kernel figures of profiles/r05_* come from rocprofv3 runs of this same script:

    python tools/reduce_bench.py [rays] [--masked 0.33] [--reps 20] [--jobs 10]

`--masked f`: a fraction f of the slots is dead (evenly spread pairs, like a mask's shadow)."""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("rays", nargs="?", type=float, default=1e7)
    ap.add_argument("--masked", type=float, default=0.0)
    ap.add_argument("--tail", type=float, default=0.0, help="the LAST fraction of the slots is dead (the shadow of C3's mask)")
    ap.add_argument("--reps", type=int, default=20)
    ap.add_argument("--jobs", type=int, default=10)
    args = ap.parse_args()
    import torch
    import __graft_entry__
    __graft_entry__.ensure_built()
    import bench
    from attosecondraytracing_amd import _lib, _abi, analysis
    from attosecondraytracing_amd.bundle import RayBundle
    import ART.ModuleDetector as mdet
    be = _lib.get_backend()
    n = int(args.rays)
    src = bench.device_source(n, 0, n, be, ("point", 0.02))
    src.intensity = torch.rand(n, dtype=torch.float64, device=be.device) + 0.5
    # a bundle "after an element": origins a metre downstream, a path, some slots dead
    b = RayBundle.allocate(n, like=src, backend=be)
    b.data.copy_(src.data)
    b.data[0:3] += 600.0 * src.data[3:6]
    b.data[6] = 600.0
    b.alive.fill_(1)
    if args.masked > 0:
        k = torch.arange(n, device=be.device)
        b.alive[((k // 2) % 1000) < int(1000 * args.masked)] = 0
    if args.tail > 0:
        b.alive[int(n * (1.0 - args.tail)):] = 0
    live = int(b.alive.sum().item())
    det = mdet.Detector(np.zeros(3))
    det.autoplace(b, 100.0)
    X, Y, O = be.empty(n), be.empty(n), be.empty(n)
    be.detector_readout(det._desc(), b.view(), src.intensity, n, XY=(X, Y), opl=O, to_host=False)
    send = torch.empty(be.survivor_bytes(n, False), dtype=torch.uint8, device=be.device)
    # the jobs of a list analysis: DIFFERENT bundles (copies here) that share one intensity array, like the chains of a loop
    # list -- the same bundle ten times would come from the caches
    others = [b] + [b.copy() for _ in range(args.jobs - 1)]
    for o in others[1:]:
        o.intensity = src.intensity
    jobs = [analysis._job(o, _abi.ART_JOB_AUTOPLACE, 100.0) for o in others]

    def timed(fn):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(args.reps):
            fn()
        e1.record()
        e1.synchronize()
        return e0.elapsed_time(e1) / args.reps * 1e3     # us

    rows = [
        ("art_bundle_sums (w)", lambda: be.fn["art_bundle_sums"](b.view(), src.intensity.data_ptr(), n, be._red_scratch().data_ptr(), be.empty(8).data_ptr(), be.stream_ptr()), 57.0 * n),
        ("art_bundle_sums (no w)", lambda: be.fn["art_bundle_sums"](b.view(), None, n, be._red_scratch().data_ptr(), be.empty(8).data_ptr(), be.stream_ptr()), 49.0 * n),
        ("art_detector_stats (w)", lambda: be.detector_stats(b.alive, X, Y, O, src.intensity, n, to_host=False), 33.0 * n),
        ("art_detector_moments (w)", lambda: be.fn["art_detector_moments"](b.alive.data_ptr(), X.data_ptr(), Y.data_ptr(), O.data_ptr(), src.intensity.data_ptr(), n, 0.0, 0.0, 700.0, be._red_scratch().data_ptr(), be.empty(8).data_ptr(), be.stream_ptr()), 33.0 * n),
        ("art_detector_scan_moments (w)", lambda: be.fn["art_detector_scan_moments"](det._desc(), b.view(), src.intensity.data_ptr(), n, 700.0, 0.0, be._red_scratch().data_ptr(), be.empty(33).data_ptr(), be.stream_ptr()), 65.0 * n),
        ("art_detector_readout (w, X Y opl)", lambda: be.detector_readout(det._desc(), b.view(), src.intensity, n, XY=(X, Y), opl=O, to_host=False), 65.0 * n + 24.0 * n),
        ("art_analyse_bundles x1 (sums + moments)", lambda: be.analyse_bundles(jobs[:1], n), 2 * 65.0 * n),
        (f"art_analyse_bundles x{args.jobs} (sums + moments)", lambda: be.analyse_bundles(jobs, n), 2 * 65.0 * n * args.jobs),
        ("art_compact", lambda: be.fn["art_compact"](b.alive.data_ptr(), n, be.scratch("compact", be.fn["art_compact_scratch_ints"](n), torch.int32).data_ptr(), be.scratch("idx", n, torch.int64).data_ptr(), be.scratch("cnt", 1, torch.int64).data_ptr(), be.stream_ptr()), 2.0 * n + 8.0 * live),
        ("art_pack_survivors", lambda: be.pack_survivors(b.alive, X, Y, O, None, 0, 1, send), 2.0 * n + 2 * 24.0 * live + (0 if live == n else 4.0 * live)),
        ("art_gaussian_intensity", lambda: be.gaussian_intensity(src.view(), [1.0, 0.0, 0.0], 1 / np.e ** 2, n), 2 * 49.0 * n + 8.0 * n),
    ]
    print(f"# {n} rays, {live} alive, {args.reps} calls each; bytes = compulsory bytes of the call")
    print(f"{'entry point':52s} {'us/call':>9s} {'MB':>9s} {'GB/s':>8s} {'of 8 TB/s':>9s}")
    for name, fn, nbytes in rows:
        us = timed(fn)
        print(f"{name:52s} {us:9.1f} {nbytes / 1e6:9.1f} {nbytes / us / 1e3:8.0f} {nbytes / us / 1e3 / 8000:9.3f}", flush=True)


if __name__ == "__main__":
    main()
