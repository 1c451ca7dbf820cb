#!/usr/bin/env python3
"""Condense the rocprofv3 CSVs written by tools/prof.sh into a small summary under profiles/.

usage: tools/summarize_profile.py gpurun_out/prof_<tag> profiles/<name>.md [rays_per_gpu]

HBM traffic follows MI355X_MICROARCH.md (HBM / rocprofv3 section): FETCH_SIZE and WRITE_SIZE come from separate
--pmc passes, are in KiB, and on gfx950 FETCH_SIZE counts 64 B per 128-B request for coalesced streaming reads,
i.e. exactly half the bytes.  Both are calibrated here on kernels of the same run whose byte counts are known
exactly (same 8-B-per-lane access pattern): k_bundle_sums_partial reads 49 B/ray and k_make_source writes 65 B/ray.
"""
import collections
import csv
import glob
import json
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
from source_hash import source_hash  # noqa: E402


def code_object_registers():
    """{kernel: (next_free_vgpr, next_free_sgpr, LDS bytes, scratch bytes)} from the compiler's metadata of the tree's
    sources (hipcc -S, no GPU needed).  rocprofv3's VGPR_Count column is NOT the allocation (it printed 48 for the
    92-VGPR fused kernel): the tables below quote the code object."""
    src = os.path.join(ROOT, "attosecondraytracing_amd", "csrc", "art_kernels.hip")
    try:
        with tempfile.TemporaryDirectory() as td:
            out = os.path.join(td, "art.s")
            subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-S", "--cuda-device-only", "-o", out, src],
                                  stderr=subprocess.DEVNULL)
            s = open(out).read()
    except Exception:
        return {}
    res = {}
    for m in re.finditer(r"\.amdhsa_kernel (\S+)(.*?)\.end_amdhsa_kernel", s, re.S):
        g = lambda key: int(re.search(r"\.amdhsa_%s (\d+)" % key, m.group(2)).group(1))
        dem = subprocess.run(["c++filt", m.group(1)], capture_output=True, text=True).stdout.strip()
        res[short(dem)] = (g("next_free_vgpr"), g("next_free_sgpr"), g("group_segment_fixed_size"), g("private_segment_fixed_size"))
    return res


def short(name):
    name = name.replace("(anonymous namespace)::", "").replace("void ", "")
    return name.split("(")[0]


def grid_threads(r):
    """Threads of a dispatch in the kernel trace: all three grid dimensions (a chain-interleaved scene launch has its tiles
    in y: its x alone is chains x 256)."""
    return int(r["Grid_Size_X"]) * int(r.get("Grid_Size_Y", 1) or 1) * int(r.get("Grid_Size_Z", 1) or 1)


def profiled_hash(src):
    """The hash tools/prof.sh recorded on the GPU box beside the passes (the build that ran); the tree's otherwise."""
    f = os.path.join(src, "source_hash.txt")
    return open(f).read().strip() if os.path.exists(f) else source_hash()


def main():
    src, dst = sys.argv[1], sys.argv[2]
    n = int(sys.argv[3]) if len(sys.argv) > 3 else 10_000_000
    extra = sys.argv[4] if len(sys.argv) > 4 else "--steps 20 --warmup 5"
    out = ["# rocprofv3 summary: " + os.path.basename(src), "",
           "commands (tools/prof.sh; three separate runs of the same bench line, profiled runs are not the headline):", "```",
           "rocprofv3 --kernel-trace --stats --output-format csv -d <out>/trace -- python3 bench.py --cpu-sample 0 " + extra,
           "rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d <out>/pmc_fetch -- python3 bench.py --cpu-sample 0 " + extra,
           "rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d <out>/pmc_write -- python3 bench.py --cpu-sample 0 " + extra,
           "```", ""]
    bj = os.path.join(src, "bench_trace.json")
    if os.path.exists(bj):
        try:
            b = json.loads(open(bj).read().strip().splitlines()[-1])
            out += ["bench line of the kernel-trace pass (profiled run, not the headline number):", "```",
                    json.dumps({k: b[k] for k in ("value", "ms_per_step", "config", "roofline")}), "```", ""]
        except Exception:
            pass
    # ---- kernel trace: only the full-size launches (largest grid per kernel)
    kt = glob.glob(os.path.join(src, "trace", "*", "*_kernel_trace.csv"))[0]
    rows = list(csv.DictReader(open(kt)))
    byk = collections.defaultdict(list)
    for r in rows:
        byk[short(r["Kernel_Name"])].append((grid_threads(r), int(r["End_Timestamp"]) - int(r["Start_Timestamp"]),
                                            r.get("VGPR_Count", ""), r.get("SGPR_Count", ""), r.get("LDS_Block_Size", "")))
    regs = code_object_registers()
    out += ["## kernel trace (`rocprofv3 --kernel-trace --stats`), full-size launches only", "",
            "(VGPR / SGPR / LDS / scratch: `next_free_vgpr` etc. of the code object built from these sources, source hash "
            + profiled_hash(src) + "; rocprofv3's own VGPR_Count column is not the allocation)", "",
            "| kernel | launches | avg us | min us | max us | grid threads | VGPR | SGPR | LDS B | scratch B |", "|---|---:|---:|---:|---:|---:|---:|---:|---:|---:|"]
    tot = 0
    stat = {}
    for k, v in sorted(byk.items(), key=lambda kv: -sum(x[1] for x in kv[1])):
        g = max(x[0] for x in v)
        big = [x for x in v if x[0] == g]
        d = [x[1] for x in big]
        stat[k] = (len(d), sum(d) / len(d))
        if sum(d) < 20000:
            continue
        rg = regs.get(k, ("?", "?", big[0][4], "?"))
        out.append(f"| {k} | {len(d)} | {sum(d)/len(d)/1e3:.1f} | {min(d)/1e3:.1f} | {max(d)/1e3:.1f} | {g} | {rg[0]} | {rg[1]} | {rg[2]} | {rg[3]} |")
    out.append("")
    timed_stat = {}
    # the timed region: bench.py reports which of its full-size fused-kernel launches lie inside it
    # (roofline.timed_region_launches, in issue order); cut the trace to them
    try:
        bl = json.loads(open(bj).read().strip().splitlines()[-1])
        lo, hi = bl["roofline"]["timed_region_launches"]
        fused = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]) - int(r["Start_Timestamp"]), grid_threads(r))
                        for r in rows if short(r["Kernel_Name"]).startswith(("k_trace_chain", "k_trace_scene"))), key=lambda t: t[0])
        gmax = max(t[2] for t in fused)
        full = [t for t in fused if t[2] >= gmax // 2]     # full-size launches (the two-ray body's grid is half a one-ray grid)
        cut = [t[1] for t in full[lo:hi]]
        if cut and hi <= len(full):
            # per kernel name: average duration of ITS launches inside the timed region (what the byte counts below, cut to
            # the same launches of their own passes, are divided by)
            named = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]) - int(r["Start_Timestamp"]), grid_threads(r),
                            short(r["Kernel_Name"])) for r in rows if short(r["Kernel_Name"]).startswith(("k_trace_chain", "k_trace_scene"))),
                           key=lambda t: t[0])
            named = [t for t in named if t[2] >= gmax // 2][lo:hi]
            for k in {t[3] for t in named}:
                d = [t[1] for t in named if t[3] == k]
                timed_stat[k] = (len(d), sum(d) / len(d))
            out.append(f"timed region of this pass (launches {lo}..{hi - 1} of {len(full)} full-size fused launches, "
                       f"`roofline.timed_region_launches`): average {sum(cut)/len(cut)/1e3:.1f} us, first {cut[0]/1e3:.1f}, "
                       f"max {max(cut)/1e3:.1f}, last {cut[-1]/1e3:.1f} us -- the clock ramp of the driver's protocol (DESIGN.md 5); "
                       f"bench.py's own `kernel_ms` of this pass: {bl['roofline']['kernel_ms']*1e3:.1f} us (events right after the "
                       f"region), `kernel_ms_sustained` {bl['roofline'].get('kernel_ms_sustained', 0)*1e3:.1f} us; average of ALL "
                       f"full-size launches of the pass: {sum(t[1] for t in full)/len(full)/1e3:.1f} us")
    except Exception as e:     # noqa: BLE001 -- older bench lines carry no launch indices
        out.append(f"(no timed-region cut: {e!r})")
    # ---- PMC passes
    def pmc(tag, counter):
        f = glob.glob(os.path.join(src, "pmc_" + tag, "*", "*_counter_collection.csv"))
        if not f:
            return {}
        acc = collections.defaultdict(list)
        fused = []
        for r in csv.DictReader(open(f[0])):
            if r["Counter_Name"] == counter:
                k = short(r["Kernel_Name"])
                acc[k].append((int(r["Grid_Size"]), float(r["Counter_Value"])))
                if k.startswith(("k_trace_chain", "k_trace_scene")):
                    fused.append((int(r["Dispatch_Id"]), int(r["Grid_Size"]), k, float(r["Counter_Value"])))
        res = {}
        for k, v in acc.items():
            g = max(x[0] for x in v)
            big = [x[1] for x in v if x[0] == g]
            res[k] = sum(big) / len(big) * 1024.0   # KiB -> bytes
        # The fused kernel serves more than one workload in a bench run (the full-history steps, the lazy-history steps, the
        # candidates of the placement look): its bytes per launch are those of the launches INSIDE THE TIMED REGION of this
        # very pass (roofline.timed_region_launches of the pass's own bench line, full-size fused launches in issue order)
        try:
            bl = json.loads(open(os.path.join(src, "bench_" + tag + ".json")).read().strip().splitlines()[-1])
            lo, hi = bl["roofline"]["timed_region_launches"]
            fused.sort()
            gmax = max(t[1] for t in fused)
            full = [t for t in fused if t[1] >= gmax // 2]
            cut = full[lo:hi]
            if cut and hi <= len(full):
                for k in {t[2] for t in cut}:
                    vals = [t[3] for t in cut if t[2] == k]
                    res[k] = sum(vals) / len(vals) * 1024.0
                cut_note[tag] = f"{tag}: launches {lo}..{hi - 1} of {len(full)}"
        except Exception as e:     # noqa: BLE001
            cut_note[tag] = f"{tag}: no timed-region cut ({e!r})"
        return res
    cut_note = {}
    fe, wr = pmc("fetch", "FETCH_SIZE"), pmc("write", "WRITE_SIZE")
    out += ["## HBM traffic per launch (`--pmc FETCH_SIZE` and `--pmc WRITE_SIZE`, separate passes)", "",
            "(fused kernels: averaged over the launches inside the timed region of each pass -- " + "; ".join(cut_note.values()) + ")", ""]
    # (bench.py launches k_bundle_sums_partial once on its all-alive source bundle for exactly this calibration)
    ksum = "k_bundle_sums_partial<false>" if "k_bundle_sums_partial<false>" in fe else "k_bundle_sums_partial"     # (a template since round 5)
    cal_r = fe.get(ksum, 0) / (49.0 * n) if fe.get(ksum) else None
    cal_w = wr.get("k_make_source", 0) / (65.0 * n) if wr.get("k_make_source") else None
    out.append(f"calibration on known byte counts ({n} rays): FETCH_SIZE/true read bytes = "
               f"{cal_r if cal_r is None else round(cal_r, 4)} (k_bundle_sums_partial, 49 B/ray; guide says 0.5 on gfx950), "
               f"WRITE_SIZE/true written bytes = {cal_w if cal_w is None else round(cal_w, 4)} (k_make_source, 65 B/ray).")
    out += ["", "Durations and bytes of one row come from the SAME launches: for the fused kernels those inside the timed region of "
            "each pass (`avg us, timed region`; `HBM GB/s` and `frac` = total MB / that time / 8000 GB/s -- the figure bench.py's "
            "`roofline` is to be compared with); the average over ALL launches of the trace pass (lazy-history steps, warm-up, "
            "event-bracketed passes included) is kept beside it for reference only.", "",
            "| kernel | FETCH_SIZE raw MB | read MB (x2 gfx950 correction) | WRITE_SIZE MB | total MB | avg us, timed region | HBM GB/s | frac of 8 TB/s | avg us, all launches |",
            "|---|---:|---:|---:|---:|---:|---:|---:|---:|"]
    traffic = {}
    for k in fe:
        if k not in stat or fe[k] + wr.get(k, 0) < 5e6:
            continue
        rd = 2.0 * fe[k]
        w = wr.get(k, 0.0)
        us_all = stat[k][1] / 1e3
        us = timed_stat[k][1] / 1e3 if k in timed_stat else us_all
        gbs = (rd + w) / us / 1e3
        out.append(f"| {k} | {fe[k]/1e6:.1f} | {rd/1e6:.1f} | {w/1e6:.1f} | {(rd+w)/1e6:.1f} | {us:.1f} | {gbs:.0f} | {gbs/8000.0:.3f} | {us_all:.1f} |")
        traffic[k] = {"read_bytes": rd, "write_bytes": w, "total_bytes": rd + w, "avg_us_timed_region": us,
                      "hbm_gbs_timed_region": gbs, "frac_timed_region": gbs / 8000.0, "avg_us_all_launches": us_all,
                      "cut_to_timed_region": k in timed_stat}
    out.append("")
    box = None
    bs = os.path.join(src, "box_state.txt")
    if os.path.exists(bs):
        out += ["## box state (tools/box_state.sh, before the passes)", "", "```"] + [l.rstrip() for l in open(bs) if re.search(
            r"clock level|Power|Partition|MAX_CLK|SOCKET_POWER|MEM_|mclk|fclk|IFWI|VERSION: 0", l)][:40] + ["```", ""]
        box = os.path.basename(bs)
    json.dump({"source": os.path.basename(src), "rays_per_gpu": n, "fetch_calibration": cal_r, "write_calibration": cal_w,
               "source_hash": profiled_hash(src), "source_hash_files": "csrc/art_device.h, art_kernels.hip, art_scene.h, include/art_hip.h",
               "box_state": box, "per_launch": traffic}, open(os.path.splitext(dst)[0] + ".json", "w"), indent=1)
    open(dst, "w").write("\n".join(out) + "\n")
    print("\n".join(out))


if __name__ == "__main__":
    main()
