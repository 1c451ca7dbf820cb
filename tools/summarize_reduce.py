#!/usr/bin/env python3
"""Per-kernel table of a tools/prof_reduce.sh run (rocprofv3 kernel trace of tools/reduce_bench.py, + the FETCH_SIZE /
WRITE_SIZE passes when they were taken): launches, average / min duration, and the HBM rate the kernel's COMPULSORY
bytes give (the model below: what the kernel must read of the bundle + what it writes, per launch at `rays` slots;
counted bytes beside it when the PMC passes exist; FETCH_SIZE x 2 on gfx950 as in tools/summarize_profile.py).

    python tools/summarize_reduce.py gpurun_out/prof_<tag> [rays] [alive]  > profiles/r05_reductions.md"""
import collections
import csv
import glob
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from summarize_profile import short, grid_threads, code_object_registers  # noqa: E402


def model_bytes(n, live, jobs):
    """kernel -> compulsory bytes per launch (n slots, `live` alive)"""
    return {
        "k_bundle_sums_partial<true>": 57.0 * n, "k_bundle_sums_partial<false>": 49.0 * n,
        "k_stats_partial<true>": 33.0 * n, "k_stats_partial<false>": 25.0 * n,
        "k_moments_partial<true>": 33.0 * n, "k_moments_partial<false>": 25.0 * n,
        "k_scan_moments_partial<true>": 65.0 * n, "k_scan_moments_partial<false>": 57.0 * n,
        "k_detector_readout": 65.0 * n + 24.0 * n,
        "k_analysis_sums": 65.0 * n, "k_analysis_moments": 65.0 * n,
        "k_compact_count": 1.0 * n, "k_compact_scatter": 1.0 * n + 8.0 * live,
        "k_survivor_scatter": 1.0 * n + 2 * 24.0 * live + (0.0 if live == n else 4.0 * live),
        "k_gauss_max_partial": 49.0 * n, "k_gauss_weights": 49.0 * n + 8.0 * n,
        "k_make_source": 65.0 * n,
    }


def read_trace(d):
    rows = []
    for f in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
        rows += list(csv.DictReader(open(f)))
    return rows


def read_pmc(d, counter):
    per = collections.defaultdict(list)
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if r.get("Counter_Name") == counter:
                per[(short(r["Kernel_Name"]), int(r["Grid_Size"]) if r.get("Grid_Size") else 0)].append(float(r["Counter_Value"]))
    return per


def main():
    src = sys.argv[1]
    n = int(float(sys.argv[2])) if len(sys.argv) > 2 else 10_000_000
    live = int(float(sys.argv[3])) if len(sys.argv) > 3 else n
    rows = read_trace(os.path.join(src, "trace"))
    regs = code_object_registers()
    by = collections.defaultdict(list)
    for r in rows:
        name = short(r["Kernel_Name"])
        by[(name, grid_threads(r))].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    fetch = read_pmc(os.path.join(src, "pmc_fetch"), "FETCH_SIZE")
    write = read_pmc(os.path.join(src, "pmc_write"), "WRITE_SIZE")
    fsum, wsum = collections.defaultdict(list), collections.defaultdict(list)
    for (k, g), v in fetch.items():
        fsum[k] += v
    for (k, g), v in write.items():
        wsum[k] += v
    model = model_bytes(n, live, 1)
    print(f"# reduction / compaction kernels: rocprofv3 kernel trace of `tools/reduce_bench.py {n}` ({live} alive)\n")
    hb = os.path.join(src, "source_hash.txt")
    if os.path.exists(hb):
        print(f"build (source hash): {open(hb).read().strip()}\n")
    bt = os.path.join(src, "bench_trace.txt")
    if os.path.exists(bt):
        print("the script's own event timings of the same (profiled) run, whole entry points:\n```\n" + open(bt).read().strip() + "\n```\n")
    print("Per kernel (full-size launches: the largest grid of each kernel), `model MB` = compulsory bytes per launch; "
          "`counted MB` = FETCH_SIZE x 2 (gfx950) + WRITE_SIZE from the separate --pmc passes, where taken.\n")
    print("| kernel | launches | avg us | min us | VGPR | model MB | GB/s (avg) | frac of 8 TB/s | counted MB |")
    print("|---|---:|---:|---:|---:|---:|---:|---:|---:|")
    # rows: every kernel at its largest grid; the analysis kernels (grid.y = job) once per job count
    gy = collections.defaultdict(set)
    # (k_analysis_moments runs a 1-D XCD-grouped grid of P8 x jobs workgroups, P8 = min(tiles, 1024) rounded up to 8)
    p8 = (min((n + 255) // 256, 1024) + 7) // 8 * 8
    for r in rows:
        nm, g = short(r["Kernel_Name"]), grid_threads(r)
        y = int(r.get("Grid_Size_Y", 1) or 1)
        if nm == "k_analysis_moments" and y == 1:
            y = max(1, g // (p8 * 256))
        gy[nm].add((g, y))
    names = sorted({k for k, _ in by if k.startswith("k_")})
    for k in names:
        grids = sorted(gy[k])
        per_job = k in ("k_analysis_sums", "k_analysis_moments")
        picks = grids if per_job else [max(grids)]
        for g, y in picks:
            if per_job and g < max(gg for gg, yy in grids if yy == y):
                continue                    # (smaller bundles of the warm-up)
            t = by[(k, g)]
            avg, mn = sum(t) / len(t), min(t)
            mb = model.get(k)
            if mb is not None and per_job:
                mb *= y
            counted = None
            if (fsum.get(k) or wsum.get(k)) and not per_job:
                fv = sorted(fsum.get(k, [0.0]))[-len(t):]
                wv = sorted(wsum.get(k, [0.0]))[-len(t):]
                counted = (2.0 * sum(fv) / max(len(fv), 1) + sum(wv) / max(len(wv), 1)) * 1024 / 1e6
            vg = regs.get(k, ("?",))[0]
            label = k + (f" ({y} job{'s' if y > 1 else ''})" if per_job else "")
            if mb is None:
                print(f"| {label} | {len(t)} | {avg:.1f} | {mn:.1f} | {vg} | - | - | - | {'' if counted is None else '%.1f' % counted} |")
            else:
                print(f"| {label} | {len(t)} | {avg:.1f} | {mn:.1f} | {vg} | {mb / 1e6:.1f} | {mb / avg / 1e3:.0f} | {mb / avg / 1e3 / 8000:.3f} | "
                      f"{'' if counted is None else '%.1f' % counted} |")
    bs = os.path.join(src, "box_state.txt")
    if os.path.exists(bs):
        print("\n## box state\n\n```\n" + open(bs).read().strip()[:1500] + "\n```")


if __name__ == "__main__":
    main()
