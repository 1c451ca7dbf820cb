"""bench.py's parts: launcher (no GPU, no torch), workloads (the five BASELINE.json configurations), baselines (the CPU
figures and the parity of the sample), roofline (byte models, committed PMC profiles), worker (the timed regions and the
JSON line).  `bench.py` at the repo root is the entry point the driver calls."""
