"""Empty stand-in: the reference imports `colorcet` at module scope (ART/ModuleAnalysisAndPlots.py:17-19)
but only uses it inside plotting functions, which the golden generator never calls."""
