#!/usr/bin/env python3
"""Regenerate every golden fixture from the reference into a scratch directory and compare with the committed files:
all arrays must be identical, bit for bit (the scene JSON may differ in its timing field only).  Build container only.

    python tests/golden/check_regeneration.py [scratch_dir]
"""
import glob
import json
import os
import subprocess
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))


def main():
    out = sys.argv[1] if len(sys.argv) > 1 else tempfile.mkdtemp(prefix="golden_regen_")
    os.makedirs(out, exist_ok=True)
    env = dict(os.environ, ART_GOLDEN_OUT=out, PYTHONDONTWRITEBYTECODE="1")
    for script in ("generate_goldens.py", "generate_render_goldens.py"):
        subprocess.check_call([sys.executable, os.path.join(HERE, script)], env=env, stdout=subprocess.DEVNULL, cwd=HERE)
    bad, n = [], 0
    for f in sorted(glob.glob(os.path.join(HERE, "*.npz"))):
        g = os.path.join(out, os.path.basename(f))
        if not os.path.exists(g):
            bad.append((os.path.basename(f), "not regenerated"))
            continue
        a, b = np.load(f), np.load(g)
        if set(a.files) != set(b.files):
            bad.append((os.path.basename(f), "different keys"))
            continue
        for k in a.files:
            if k == "scene_json":
                x, y = json.loads(str(a[k])), json.loads(str(b[k]))
                x.pop("reference_trace_seconds", None)
                y.pop("reference_trace_seconds", None)
                same = x == y
            else:
                same = a[k].shape == b[k].shape and np.array_equal(a[k], b[k], equal_nan=a[k].dtype.kind in "fc")
            if not same:
                bad.append((os.path.basename(f), k))
        n += 1
    print(f"{n} fixtures compared, {len(bad)} differences")
    for item in bad:
        print("  ", item)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
