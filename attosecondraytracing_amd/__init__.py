"""attosecondraytracing_amd -- MI355X-native ray-bundle propagation behind ART's Python API.

The hot path (ART/ModuleProcessing.py:250-313 `RayTracingCalculation` + the detector read-out) runs in
hand-written HIP kernels for gfx950 (csrc/, C ABI in include/art_hip.h); the modules here mirror the
reference's module names so that `import ART.ModuleMirror as mmirror` etc. keep working through the `ART`
alias package at the repository root."""
__version__ = "0.93-mi355x.1"
