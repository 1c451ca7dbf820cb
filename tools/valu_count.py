#!/usr/bin/env python3
"""Static instruction counts per basic block of one kernel of csrc/art_kernels.hip (no GPU needed):
    python tools/valu_count.py [-D...] --kernel 'k_trace_element<3, false>'
Prints VALU / transcendental / SALU / memory counts per block and the totals -- a proxy for the dynamic VALU count that
bounds the tracing kernels (profiles/r02_relay4_sq.md); blocks of loops are marked."""
import collections
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "attosecondraytracing_amd", "csrc", "art_kernels.hip")


def kernel_blocks(asm, want):
    parts = re.split(r"\n(_Z\w+):[^\n]*\n", asm)
    for name, body in zip(parts[1::2], parts[2::2]):
        dem = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
        dem = dem.replace("(anonymous namespace)::", "").split("(")[0].replace("void ", "")
        if dem != want or "s_endpgm" not in body:
            continue
        code = body[:body.rfind("s_endpgm") + 8]
        blocks, cur, loop = [], ["entry", [], False], False
        for l in code.split("\n"):
            m = re.match(r"^(\.LBB\w+):(.*)", l)
            if m:
                blocks.append(cur)
                cur = [m.group(1), [], "Loop" in m.group(2)]
            elif l.startswith("\t"):
                t = l.strip()
                if t and not t.startswith((".", ";")):
                    cur[1].append(t.split()[0])
        blocks.append(cur)
        return blocks
    raise SystemExit("kernel not found: " + want)


def main():
    args = sys.argv[1:]
    want = "k_trace_element<3, false>"
    if "--kernel" in args:
        k = args.index("--kernel")
        want = args[k + 1]
        del args[k:k + 2]
    with tempfile.TemporaryDirectory() as td:
        out = os.path.join(td, "art.s")
        subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-S", "--cuda-device-only", "-o", out,
                               SRC] + args, stderr=subprocess.DEVNULL)
        asm = open(out).read()
    tot = collections.Counter()
    print(f"{'block':14s} {'n':>5s} {'valu':>5s} {'f64':>5s} {'trans':>5s} {'mov':>4s} {'cndm':>4s} {'salu':>5s} {'mem':>4s}")
    for name, ins, loop in kernel_blocks(asm, want):
        if not ins:
            continue
        c = collections.Counter()
        for i in ins:
            c["n"] += 1
            if i.startswith("v_"):
                c["valu"] += 1
                c["f64"] += i.endswith("_f64") or "_f64_" in i
                c["trans"] += bool(re.match(r"v_(rcp|rsq|sqrt)_", i))
                c["mov"] += i.startswith("v_mov")
                c["cndm"] += i.startswith("v_cndmask")
            elif i.startswith("s_"):
                c["salu"] += 1
            else:
                c["mem"] += 1
        tot.update(c)
        print(f"{name:14s} {c['n']:5d} {c['valu']:5d} {c['f64']:5d} {c['trans']:5d} {c['mov']:4d} {c['cndm']:4d} {c['salu']:5d} {c['mem']:4d}"
              + ("  (loop)" if loop else ""))
    print(f"{'total':14s} {tot['n']:5d} {tot['valu']:5d} {tot['f64']:5d} {tot['trans']:5d} {tot['mov']:4d} {tot['cndm']:4d} {tot['salu']:5d} {tot['mem']:4d}")


if __name__ == "__main__":
    main()
