"""dehaze_hip - MI355X (gfx950) kernels + host glue for the Uformer_ProbSparse training path.

Importing this package loads libdehaze_hip.so (hand-written HIP kernels behind the C-ABI of
include/dehaze_hip.h) and fails loudly if it has not been built: there is no PyTorch/CPU fallback.
"""
from . import _lib

_lib.load()

from . import ops  # noqa: E402,F401
