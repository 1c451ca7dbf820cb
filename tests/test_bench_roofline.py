"""CPU: the byte models behind bench.py's `roofline` object (tools/bench/roofline.py) -- what a fused chain launch must
move, stated in DESIGN.md 5: per chain n (57 + 8 w) read once, per element 64 B per live slot + the alive byte of every
slot, 24 B per read-out ray (+ 176 B of partial statistics per workgroup in the compulsory figure); a scene whose chains
share their source reads it once."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from tools.bench import roofline as rl  # noqa: E402


def test_fused_chain_bytes_relay4():
    n = 10_000_000
    algo, comp = rl.chain_bytes([n] * 4, n, True, True)
    assert algo == n * 65.0 + 4 * 65.0 * n + 24.0 * n                # 3.49 GB: read once, four bundles, the read-out
    assert comp == algo + 176.0 * ((n + 255) // 256)
    # what SURVEY 8(d) prices for the same step (one kernel per element, the ray re-read): more than the fused launch moves
    assert rl.SURVEY_BYTES_PER_INTERSECTION * 4 * n + rl.SURVEY_BYTES_PER_READOUT_RAY * n > algo


def test_dead_slots_write_their_alive_byte_only():
    n = 1000
    algo, _ = rl.chain_bytes([600, 500], n, False, False)
    assert algo == 57.0 * n + (64.0 * 600 + n) + (64.0 * 500 + n)
    per_element, _ = rl.chain_bytes([600, 500], n, False, False, fused_kernels=False)
    assert per_element == algo + 57.0 * n                              # one launch per element re-reads the bundle


def test_shared_source_is_counted_once():
    n, chains = 1_000_000, 11
    assert rl.shared_source_credit(n, chains) == 57.0 * n * 10
    assert rl.shared_source_credit(n, 1) == 0.0
    total = sum(rl.chain_bytes([n // 2, n // 2, n // 2], n, True, True)[0] for _ in range(chains)) - rl.shared_source_credit(n, chains)
    one = rl.chain_bytes([n // 2, n // 2, n // 2], n, True, True)[0]
    assert total == chains * one - 10 * 57.0 * n and total < chains * one


def test_peaks_are_the_guides():
    assert rl.HBM_PEAK_GBS == 8000.0 and rl.XGMI_LINK_GBS == 153.0
