"""MI355X-native deflated-MLMC Hutchinson trace engine for the 2-D Schwinger operator.

The sub-modules mirror the reference's files (gateway, examples, matrix, multigrid,
stoch_trace, utils); the arithmetic of the hot path lives in ``libschwinger_hip.so``
(``csrc/``, C ABI in ``include/schwinger_hip.h``) and is reached through ``engine``.
"""
__all__ = ["engine", "hierarchy", "matrix", "multigrid", "utils", "stoch_trace", "examples",
           "gateway", "dist"]
__version__ = "0.1.0"
