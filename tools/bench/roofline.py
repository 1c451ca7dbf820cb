"""Byte models and the committed PMC profiles behind bench.py's `roofline` object."""
import json
import os
import re
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def log(*a):
    print(*a, file=sys.stderr, flush=True)

HBM_PEAK_GBS = 8000.0                 # MI355X HBM3E spec peak (MI355X_MICROARCH.md; ~6300 GB/s is what a copy achieves)
XGMI_LINK_GBS = 153.0                 # one xGMI link (7 per GPU, point to point)
# SURVEY.md 8(d) prices an intersection at 128 B (read 48 + 8 + 4, write 48 + 8 + 8 + 4) and a read-out ray at 88 B: a model of
# ONE KERNEL PER ELEMENT, each re-reading the ray.  The fused chain reads a ray ONCE per chain, so its algorithmic bytes are
#   per chain:  n (57 + 8 w)  read   (7 fp64 streams + the alive byte, + the weight with a fused read-out)
#               sum over elements of (64 live_k + n)  written   (8 fp64 streams per live slot + every slot's alive byte)
#               24 live_last   written by the fused read-out (X, Y, path)
# -- the figure `achieved_algorithmic` uses; the survey's per-element figure is kept as a label only (with the fused
# kernel it gives a "fraction" above 1: 6.0 GB per relay4 launch against 3.5 GB moved).
SURVEY_BYTES_PER_INTERSECTION = 128.0
SURVEY_BYTES_PER_READOUT_RAY = 88.0


def profiled_traffic(config, kernel_pattern, rays):
    """HBM bytes per launch of the kernel whose name matches the regular expression `kernel_pattern` (anchored at the
    start; masked chains launch the two-rays-per-lane bodies k_trace_scene2 / k_trace_chain2), from the newest committed rocprofv3
    PMC summary of this workload (profiles/rNN_<config>*.json, written by tools/summarize_profile.py from separate
    --pmc FETCH_SIZE / WRITE_SIZE passes with the gfx950 x2 read correction) THAT WAS TAKEN ON THIS BUILD: a profile whose
    `source_hash` (csrc/* + include/art_hip.h at profiling time) differs from the tree's is dropped, and the line says so.
    -> (bytes, file, kernel) or None, and a note."""
    import glob
    from tools.source_hash import source_hash
    here = source_hash()
    best, dropped = None, []
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", f"r[0-9][0-9]_{config}.json")))    # newest round last
    for f in files:
        try:
            j = json.load(open(f))
        except Exception:
            continue
        hits = [k for k in j.get("per_launch", {}) if re.match(kernel_pattern, k)]
        if j.get("rays_per_gpu") == rays and hits:
            if j.get("source_hash") != here:
                dropped.append(f"{os.path.relpath(f, ROOT)} (sources {j.get('source_hash', 'unrecorded')} != {here})")
                continue
            best = (j["per_launch"][hits[0]]["total_bytes"], os.path.relpath(f, ROOT), hits[0])
    note = None
    if best is None and dropped:
        note = "no PMC profile of THIS build: dropped " + "; ".join(dropped)
    return best, note




def shared_source_credit(n, n_chains):
    """Bytes a scene launch does NOT have to move because its chains read ONE source bundle (57 B per slot: 7 fp64 streams +
    the alive byte; the weights of a fused read-out are read per chain): the XCD-grouped grid fetches a tile of it once for
    all chains (csrc/art_kernels.hip scene_wg), so the launch's minimum is one read, not one per chain."""
    return 57.0 * n * max(0, n_chains - 1)


def chain_bytes(live, n, has_w, fused_readout, fused_kernels=True):
    """Bytes the launch(es) of one step must move for ONE chain, from the run's own survivor counts `live` (after every
    element): (algorithmic, compulsory).  Fused kernels read the source once -- 7 fp64 streams + the alive byte = 57 B per
    slot, + 8 B of weight with a fused read-out -- and write, per element, 64 B per LIVE slot (pairs of dead slots are
    dropped by the range check) + the alive byte of every slot; a fused read-out adds 24 B per surviving ray and, in the
    compulsory figure, 176 B of partial statistics per workgroup.  Per-element launches re-read every bundle."""
    if fused_kernels:
        algo = n * (57.0 + (8.0 if has_w else 0.0)) + sum(64.0 * x + n for x in live)
    else:
        algo = sum(57.0 * n + 64.0 * x + n for x in live)
    comp = algo
    if fused_readout:
        algo += 24.0 * live[-1]
        comp += 24.0 * live[-1] + 176.0 * ((n + 255) // 256)
    return algo, comp
