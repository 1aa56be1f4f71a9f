"""MI355X-native penalty-evaluation linear-solve path of FletcherPenaltySolver.jl.

Import as `fps_amd` (see /fps_amd.py: the directory name carries a dot).  Submodules:
  problems      synthetic workloads (numpy only)
  nlpmodels     NLPModels-shaped host models used by tests
  qdsolver      QDSolver seam + HIPQDSolver (libfpsq)
  penalty_nlp   FletcherPenaltyNLP (obj / grad! / objgrad!)
  device_qp     device-resident eq-QP evaluation (the benchmark's unit of work)
  fps_solve     condensed host mirror of the reference's outer loop (fps_solve entry) on the HIP back-ends
"""
__all__ = ["problems", "nlpmodels", "qdsolver", "penalty_nlp", "device_qp", "distributed", "fps_solve"]
