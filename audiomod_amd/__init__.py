"""audiomod_amd -- MI355X-native phase-vocoder engine behind the audiomod::phasevocoder interface.

Only what the hot path needs lives here: csrc/ (HIP kernels + host engine + C ABI),
engine.py (ctypes mirror of the reference interface) and signals.py (synthetic inputs).
"""
from . import engine, signals  # noqa: F401
