"""deep_cartograph_amd -- MI355X-native engine for deep_cartograph's CV-fit hot path.

Host-side mirror of the reference's calculator / clustering interface
(deep_cartograph/modules/cv_learning/cv_calculator.py, modules/statistics/statistics.py)
over hand-written HIP kernels reached through the C-ABI of ``libdcv.so`` (include/dcv.h).
There is no CPU fallback: every numeric entry point raises if the library or a GPU is
missing.
"""
__version__ = "0.1.0"
