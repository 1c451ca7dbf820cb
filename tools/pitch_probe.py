#!/usr/bin/env python3
"""Where the history rows of a fused chain launch lie in memory -- does it matter?  ONE device buffer is allocated once;
the per-element output views (8 fp64 rows per element) of the same launch are laid into it with a chosen ROW PITCH and a
chosen START OFFSET, variant after variant, alternating, every launch bracketed by HIP events: the allocation is the
same for all variants, only the addresses of the rows differ.

    python tools/pitch_probe.py [--config relay4|C4] [--rays N] [--rounds 6] [--launches 20]
                                [--pitches "0;4096;1048576;8388608"]   extra bytes per row on top of the default pitch
                                [--offsets "0;65536;2097152"]          start of the first row inside the buffer
"""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", default="relay4")
    ap.add_argument("--rays", type=int, default=0)
    ap.add_argument("--rounds", type=int, default=6)
    ap.add_argument("--launches", type=int, default=20)
    ap.add_argument("--pitches", default="0;512;4096;65536;1048576;2097152;4194304;8388608;12582912;16777216;33554432")
    ap.add_argument("--offsets", default="0")
    ap.add_argument("--knobs", default="", help="with --buffers: ';'-separated environment settings of the library (e.g. "
                    "'ART_CHAIN_RPL=1;ART_CHAIN_RPL=2'), each timed in every buffer")
    ap.add_argument("--buffers", default="", help="instead: ';'-separated EXTRA sizes in bytes -- one buffer per entry, the same "
                    "launch (default pitch, offset 0) into each of them: is it the allocation that matters?")
    args = ap.parse_args()
    import torch
    import bench
    import __graft_entry__
    __graft_entry__.ensure_built()
    from attosecondraytracing_amd import _lib, _abi
    from attosecondraytracing_amd.bundle import RayBundle
    import ART.ModuleProcessing as mp
    be = _lib.get_backend()
    if args.config == "relay4":
        els, kind = bench.build_scene(4)[0].optical_elements, ("point", 0.02)
        n = args.rays or 10_000_000
    else:
        lists, kind, _ = getattr(bench, "scene_" + args.config.lower())()
        els = lists[0]
        n = args.rays or {"C4": 12_500_000, "C5": 10_000_000}[args.config]
    src = bench.device_source(n, 0, n, be, kind)
    m = len(els)
    descs = [mp.element_descriptor(oe, args.config != "C5", be)[0] for oe in els]
    pitches = [int(v) for v in args.pitches.split(";")]
    offsets = [int(v) for v in args.offsets.split(";")]
    base_pitch = RayBundle._pitch(n) * 8                                   # bytes, the product's default
    span = max(offsets) + m * 8 * (base_pitch + max(pitches)) + 4096
    buf = torch.empty(span, dtype=torch.uint8, device=be.device)            # ONE allocation for every variant
    alive = torch.empty((m, RayBundle._pitch(n, 512)), dtype=torch.uint8, device=be.device)
    p0 = buf.data_ptr()
    vin = src.view()
    print(f"# {args.config} {n} rays x {m} elements; default row pitch {base_pitch} B; buffer {span / 2**30:.2f} GiB at "
          f"0x{p0:x} (mod 2 MiB = {p0 % (1 << 21)}), alive rows at 0x{alive.data_ptr():x}")

    def views(pitch_extra, offset):
        pitch = base_pitch + pitch_extra
        out = []
        for k in range(m):
            v = _abi.ArtBundleView()
            a = p0 + offset + k * 8 * pitch
            v.ox, v.oy, v.oz, v.dx, v.dy, v.dz = a, a + pitch, a + 2 * pitch, a + 3 * pitch, a + 4 * pitch, a + 5 * pitch
            v.path, v.incidence = a + 6 * pitch, a + 7 * pitch
            v.alive = alive[k].data_ptr()
            out.append(v)
        return out

    if args.buffers:
        extra = [int(v) for v in args.buffers.split(";")]
        bufs = [torch.empty(m * 8 * base_pitch + 4096 + e, dtype=torch.uint8, device=be.device) for e in extra]
        knobs = [k for k in args.knobs.split(";") if k] or [""]
        times = {j: [] for j in range(len(bufs))}
        ktimes = {(j, k): [] for j in range(len(bufs)) for k in knobs}
        for rnd in range(args.rounds + 1):
            for j, b in enumerate(bufs):
                p0 = b.data_ptr()
                vs = views(0, 0)
                for k in knobs:
                    for kv in k.split():
                        os.environ[kv.split("=")[0]] = kv.split("=")[1]
                    be.trace_events = []
                    for _ in range(args.launches):
                        be.trace_chain(descs, vin, vs, n)
                    torch.cuda.synchronize()
                    ev, be.trace_events = be.trace_events, None
                    for kv in k.split():
                        os.environ.pop(kv.split("=")[0], None)
                    if rnd > 0:
                        ktimes[(j, k)].append(float(np.mean([a.elapsed_time(c) for a, c in ev])))
                if rnd > 0:
                    times[j].append(ktimes[(j, knobs[0])][-1])
        # is it the memory itself?  a plain fill and a plain copy of the same bytes into each buffer (first m * 8 rows)
        nbytes = m * 8 * base_pitch
        srcbuf = torch.empty(nbytes // 2, dtype=torch.uint8, device=be.device)
        fill, copy = {}, {}
        for rnd in range(3):
            for j, b in enumerate(bufs):
                e = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
                e[0].record()
                for _ in range(5):
                    b[:nbytes].view(torch.float64).fill_(1.0)
                e[1].record()
                for _ in range(5):
                    b[:nbytes // 2].copy_(srcbuf)
                e[2].record()
                torch.cuda.synchronize()
                fill[j] = nbytes * 5 / e[0].elapsed_time(e[1]) * 1e-9
                copy[j] = nbytes * 5 / e[1].elapsed_time(e[2]) * 1e-9      # read + written bytes
        # the BARE pattern (tools/pattern_lib.hip: the same reads and writes without any arithmetic) into the same views: the
        # launch against its floor IN THE SAME MEMORY -- the only comparison the allocation lottery does not blur
        bare = {}
        lib_path = os.path.join(ROOT, "tools", "_build", "libpattern.so")
        if os.path.exists(lib_path) and m in (1, 2, 3, 4, 8):
            import ctypes as C
            pl = C.CDLL(lib_path)
            pl.pattern_launch.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int64, C.c_void_p]
            sp = be.stream_ptr()
            for rnd in range(3):
                for j, b in enumerate(bufs):
                    p0 = b.data_ptr()
                    varr = (_abi.ArtBundleView * m)(*views(0, 0))
                    for _ in range(3):
                        assert pl.pattern_launch(C.byref(vin), varr, m, n, sp) == 0
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record()
                    for _ in range(args.launches):
                        pl.pattern_launch(C.byref(vin), varr, m, n, sp)
                    e1.record()
                    e1.synchronize()
                    bare[j] = e0.elapsed_time(e1) / args.launches
        for j, b in enumerate(bufs):
            t = np.array(times[j])
            tail = f"   bare pattern {bare[j]:.4f} ms -> launch / floor {np.median(t) / bare[j]:.3f}" if j in bare else ""
            if len(knobs) > 1:
                tail += "   " + "  ".join(f"[{k}] {np.median(ktimes[(j, k)]):.4f}" for k in knobs)
            print(f"buffer {j}: {b.numel() / 2**30:.3f} GiB at 0x{b.data_ptr():x}  median {np.median(t):.4f} ms  min {t.min():.4f}  max {t.max():.4f}"
                  f"   fill {fill[j]:.2f} TB/s  copy {copy[j]:.2f} TB/s{tail}")
        return
    variants = [(pe, off) for off in offsets for pe in pitches]
    times = {v: [] for v in variants}
    for rnd in range(args.rounds + 1):
        for v in variants:
            vs = views(*v)
            be.trace_events = []
            for _ in range(args.launches):
                be.trace_chain(descs, vin, vs, n)
            torch.cuda.synchronize()
            ev, be.trace_events = be.trace_events, None
            if rnd > 0:
                times[v].append(float(np.mean([a.elapsed_time(b) for a, b in ev])))
    base = np.median(times[variants[0]])
    for (pe, off) in variants:
        t = np.array(times[(pe, off)])
        print(f"offset {off:>10d}  pitch +{pe:>10d} B  median {np.median(t):.4f} ms  min {t.min():.4f}  ratio {np.median(t) / base:.3f}")


if __name__ == "__main__":
    main()
