import glob
import json
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN_DIR = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run on the GPU box via gpurun)")


PARITY_REPORT = []


def report(line):
    """A line for the end-of-run parity summary (shown with -q too, so the driver's GPUTEST tail carries the worst
    relative errors measured on the GPU)."""
    PARITY_REPORT.append(str(line))


def pytest_terminal_summary(terminalreporter):
    if PARITY_REPORT:
        terminalreporter.write_line("---- parity summary (worst relative errors; bars: 1e-10, incidence 1e-9 rad) ----")
        for line in PARITY_REPORT:
            terminalreporter.write_line(line)


def load_golden(name):
    """Load one fixture: returns (scene dict, arrays dict)."""
    z = np.load(os.path.join(GOLDEN_DIR, name + ".npz"), allow_pickle=False)
    arrays = {k: z[k] for k in z.files}
    scene = json.loads(str(arrays.pop("scene_json"))) if "scene_json" in arrays else None
    return scene, arrays


def chain_golden_names():
    names = []
    for p in sorted(glob.glob(os.path.join(GOLDEN_DIR, "*.npz"))):
        n = os.path.basename(p)[:-4]
        if n in ("zernike_tierA", "geometry_units", "render_grids") or n.startswith("analysis_"):
            continue
        names.append(n)
    return names
