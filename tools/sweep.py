#!/usr/bin/env python3
"""Throughput sweep on one MI355X: rays N in {1e5..1e8} x mirrors M in {2,4,8} (relay of toroids) plus the BASELINE
scenes C2/C3 (loop lists of mask + 2 toroids, every chain of the list in ONE launch), C4 (8-element mixed chain) and
C5 (Zernike-deformed parabola).  Trace-only timing (HIP events around `reps` traces, after two warm-up traces of the
same kind), full per-element history, fp64.  Three ways of issuing the trace:
  chain    one fused launch per chain, eager (art_trace_chain)
  element  one launch per element, eager (art_trace_element)
  program  graph.SceneProgram: device-resident scene table, one launch for ALL chains, replayed from a HIP graph
One fraction of the 8 TB/s HBM peak per row, on the bytes the launch must move (57 B per slot read once per chain or
element launch + 65 B per live slot and element written; the PMC counters agree with this count to 0.1 %, profiles/).
(SURVEY 8d's 128 B per intersection prices an UNFUSED element step; a fused chain moves fewer bytes than that, so a
fraction on 128 B would pass 1 and is not printed -- the intersections/s column is the rate it would scale.)
Writes a markdown table."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def _survivors(outs):
    return [int(o.alive.sum().item()) for o in outs]


def time_trace(src, element_lists, mode, reps, **kw):
    """-> (ms per trace of all chains, intersections, bytes really moved, survivors of the last bundle / slots)."""
    import ART.ModuleProcessing as mp
    from attosecondraytracing_amd.graph import SceneProgram
    n, c = src.n_slots, len(element_lists)
    if mode == "program":
        prog = SceneProgram([src] * c, element_lists, **kw)
        run = prog.run
        outs = run()
    else:
        def run():
            return [mp.RayTracingCalculation(src, els, mode=mode, **kw) for els in element_lists]
        outs = run()
    inter, moved = 0, 0
    if mode == "program" and c > 1:
        moved -= 57 * n * (c - 1)      # the chains of a program share ONE source: the XCD-grouped launch reads it once
    for o in outs:
        s = _survivors(o)
        entering = [n] + s[:-1]
        inter += sum(entering)
        reads = 57 * n if mode != "element" else 57 * n * len(o)       # dead slots are read too
        moved += reads + sum(64 * k + n for k in s)                     # 64 B per survivor + 1 alive byte per slot
    surv = _survivors(outs[-1])[-1] / n
    del outs
    for _ in range(2):                # warm-up of this very (scene, mode) pair: first-launch costs stay out of the timing
        o = run()
        del o
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        o = run()
        del o
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps, inter, moved, surv


def point_source(n, div, be):
    return bench.device_source(n, 0, n, be, ("point", div))


def plane_source(n, radius, be):
    """PlaneWaveDisk emits N - 1 rays (ART/ModuleSource.py:162)."""
    return bench.device_source(n - 1, 0, n, be, ("plane", radius), 800e-6)


# scene builders kept under their round-1 names for the other tools
def scene_c3(n):
    return bench.scene_c3()[0][3], 0.025


def scene_c2(n):
    return bench.scene_c2()[0][5], 0.025


def scene_mixed8(n):
    return bench.scene_c4()[0][0], 0.03


def scene_c5(n, perturbed):
    return bench.scene_c5()[0][0]


def main():
    import gc
    torch.cuda.set_device(0)
    from attosecondraytracing_amd import _lib
    be = _lib.get_backend()
    rows = []
    bench.build_scene(2)                 # import everything first, then take the cyclic collector out of the timings:
    gc.collect()                         # a full collection stops the host for ~40 ms (one outlier row per sweep)
    gc.freeze()

    def add(name, src, element_lists, mode, reps, **kw):
        ms, inter, moved, surv = time_trace(src, element_lists, mode, reps, **kw)
        rows.append((name, src.n_slots, len(element_lists[0]), len(element_lists), mode, ms, inter / ms * 1e3,
                     moved / ms * 1e3 / 8e12, surv))
        print(rows[-1], flush=True)

    for M in (2, 4, 8):
        chain, _ = bench.build_scene(M)
        for n in (100_000, 1_000_000, 10_000_000, 100_000_000):
            if n * M * 65 > 120e9:
                continue
            src = point_source(n, 0.02, be)
            reps = max(5, min(200, int(4e8 / (n * M))))
            for mode in ("chain", "element", "program"):
                if mode == "program" and n > 10_000_000:
                    continue
                add(f"relay{M} (toroids)", src, [chain.optical_elements], mode, reps)
            del src
            torch.cuda.empty_cache()
    for name, fn, sizes in (("C2 f-x-f: 11 chains x (mask + 2 toroids)", bench.scene_c2, (100_000, 1_000_000)),
                            ("C3 twisted: 10 chains x (mask + 2 toroids)", bench.scene_c3, (1_000_000, 10_000_000))):
        lists, kind, _ = fn()
        for n in sizes:
            src = point_source(n, kind[1], be)
            for mode in ("chain", "program"):
                add(name, src, lists, mode, 10 if n >= 10_000_000 else 30)
            del src
            torch.cuda.empty_cache()
    lists, kind, _ = bench.scene_c4()
    for n in (1_000_000, 12_500_000):
        src = point_source(n, kind[1], be)
        for mode in ("chain", "element", "program"):
            add("C4 mixed8: OAP, plane, 2 toroids, 2 planes, OAP, plane", src, lists, mode, 10)
        del src
    lists, kind, _ = bench.scene_c5()
    src = plane_source(10_000_000, kind[1], be)
    for ign in (True, False):
        for mode in ("chain", "element"):
            add(f"C5 parabola + Zernike(order 6), IgnoreDefects={ign}", src, lists, mode, 10, IgnoreDefects=ign)
    out = ["| scene | rays | elements | chains | issue | ms / trace | intersections/s | frac of 8 TB/s (bytes moved) | survivors |",
           "|---|---:|---:|---:|---|---:|---:|---:|---:|"]
    for r in rows:
        out.append(f"| {r[0]} | {r[1]:.0e} | {r[2]} | {r[3]} | {r[4]} | {r[5]:.4f} | {r[6]:.3e} | {r[7]:.2f} | {r[8]:.3f} |")
    path = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "gpurun_out", "sweep.md")
    os.makedirs(os.path.dirname(path), exist_ok=True)
    open(path, "w").write("\n".join(out) + "\n")
    print("\n".join(out))


if __name__ == "__main__":
    main()
