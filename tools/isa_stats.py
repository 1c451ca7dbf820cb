#!/usr/bin/env python3
"""Register / spill / instruction-mix summary of the gfx950 device code (no GPU needed):
    python tools/isa_stats.py [extra hipcc flags, e.g. -DART_STORE_LDS4] [--filter trace]
Compiles csrc/art_kernels.hip with `hipcc -S --cuda-device-only` and prints one line per kernel."""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "attosecondraytracing_amd", "csrc", "art_kernels.hip")


def main():
    args = sys.argv[1:]
    flt = "trace"
    if "--filter" in args:
        k = args.index("--filter")
        flt = args[k + 1]
        del args[k:k + 2]
    with tempfile.TemporaryDirectory() as td:
        out = os.path.join(td, "art.s")
        subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-S", "--cuda-device-only", "-o", out,
                               SRC] + args, stderr=subprocess.DEVNULL)
        s = open(out).read()
    meta = {}
    for m in re.finditer(r"\.amdhsa_kernel (\S+)(.*?)\.end_amdhsa_kernel", s, re.S):
        b = m.group(2)
        g = lambda key: int(re.search(r"\.amdhsa_%s (\d+)" % key, b).group(1))
        meta[m.group(1)] = (g("next_free_vgpr"), g("next_free_sgpr"), g("group_segment_fixed_size"),
                            g("private_segment_fixed_size"))
    parts = re.split(r"\n(_Z\w+):[^\n]*\n", s)
    print(f"{'kernel':58s} vgpr sgpr   lds scratch | s_load gl_load ds_rd  rdlane buf_st fma64 rcp/rsq lines")
    for name, body in zip(parts[1::2], parts[2::2]):
        if name not in meta:
            continue
        dem = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
        dem = dem.replace("(anonymous namespace)::", "").split("(")[0]
        if flt not in dem:
            continue
        code = body[:body.rfind("s_endpgm") + 8]
        c = lambda pat: len(re.findall(pat, code))
        v, sg, lds, scr = meta[name]
        cols = [c(r"\bs_load"), c(r"\bglobal_load"), c(r"\bds_read"), c(r"v_readlane|v_writelane"), c(r"buffer_store"),
                c(r"v_fma_f64"), c(r"v_rcp_f64|v_rsq_f64"), code.count("\n")]
        print(f"{dem[:58]:58s} {v:4d} {sg:4d} {lds:5d} {scr:7d} | " + " ".join(f"{x:6d}" for x in cols))

if __name__ == "__main__":
    main()
