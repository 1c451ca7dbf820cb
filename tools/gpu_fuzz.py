#!/usr/bin/env python3
"""Long differential fuzz run on the GPU: tests/fuzz_common.py scenes (HIP kernels through the C ABI vs the pinned
oracle), far more seeds than the test suite runs.  usage: python tools/gpu_fuzz.py FIRST COUNT"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "tests"), ROOT]
import fuzz_common as fz  # noqa: E402


def main():
    first, count = int(sys.argv[1]), int(sys.argv[2])
    from attosecondraytracing_amd import _lib
    _lib.get_backend()                      # raises without the HIP library / a GPU
    worst, fails, hits, det, st = {}, [], 0, 0, {}
    t0 = time.time()
    for s in range(first, first + count):
        try:
            r = fz.run_differential([s], stats=st)
            hits += r["scenes_with_hits"]
            for k, v in r["worst"].items():
                worst[k] = max(worst.get(k, 0.0), v)
            if s % 4 == 0:
                det += fz.run_detector_fuzz([s])
        except Exception as e:  # noqa: BLE001
            fails.append((s, repr(e)[:300]))
        if (s - first) % 500 == 499:
            print(f"[{time.time() - t0:.0f} s] {s - first + 1} scenes, {len(fails)} failures", flush=True)
    print(f"scenes {count} (seeds {first}..{first + count - 1}), with hits {hits}, detector poses {det}, worst {worst}")
    print(f"local errors vs long-double truth (every element of every scene): {st.get('local_worst')}")
    print(f"scenes whose product-vs-oracle difference exceeded 1e-10 and were adjudicated by truth: "
          f"{len(set(st.get('adjudicated_seeds', [])))} {sorted(set(st.get('adjudicated_seeds', [])))[:40]}; worst error vs truth there: "
          f"{st.get('adjudicated_worst')}")
    print(f"failures: {len(fails)}")
    for f in fails[:40]:
        print(f)
    return 1 if fails else 0


if __name__ == "__main__":
    sys.exit(main())
