"""MI355X-native penalty-evaluation linear-solve path of FletcherPenaltySolver.jl.

Import as `fps_amd` (see /fps_amd.py: the directory name carries a dot).  Submodules are imported lazily
so that workload generators (`problems`) stay usable without the HIP library being built.
"""
__all__ = ["problems"]
