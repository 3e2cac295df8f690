"""minimod_amd -- MI355X-native `minimod freq` hot path.

Layout: csrc/ holds the hand-written gfx950 kernels and the C ABI (include/minimod_hip.h), csrc/host the C host
side (CLI, BAM/FASTA readers, batch flattening, output formatting); this Python package is only the thin ctypes
mirror of that ABI used by tests/ and bench.py.  There is no CPU fallback: the engine raises if the HIP library
or a GPU is missing.
"""
from .build import build_all, lib_path  # noqa: F401
from .engine import FreqEngine, MinimodHipError, klass_lut  # noqa: F401
