"""`ART` -- import alias so the reference's CONFIG scripts (`import ART.ModuleMirror as mmirror`,
`from ARTmain import main`) run unmodified on top of attosecondraytracing_amd."""
import importlib
import sys

_MODULES = ["ModuleGeometry", "ModuleOpticalRay", "ModuleSupport", "ModuleDefects", "ModuleMask", "ModuleMirror",
            "ModuleOpticalElement", "ModuleSource", "ModuleProcessing", "ModuleDetector", "ModuleOpticalChain",
            "ModuleAnalysisAndPlots", "DefaultOptions", "recursive_zernike_generator"]

for _m in _MODULES:
    _mod = importlib.import_module("attosecondraytracing_amd." + _m)
    sys.modules["ART." + _m] = _mod
    globals()[_m] = _mod
