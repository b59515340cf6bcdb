"""GEMM kernel selection for the library GEMMs that stay on the path (forward / backward-data of the token Linears).

hipBLASLt's default heuristic is not the best choice for this model's skinny shapes (T = 2k..524k tokens against
N, K = 32..2048) when the operands come cold from HBM.  `tunableop_gfx950.csv` holds the solution PyTorch's TunableOp
picked for each of the 66 GEMM shapes of the config-2 training step on an MI355X (tuned with a 512 MB rotating buffer so
that candidates are timed on cold operands; `tools/tune_gemms.sh` regenerates it).  Loading it changes WHICH hipBLASLt
kernel runs, not what it computes (fp32 in, fp32 accumulate).  The file carries validator lines (PyTorch / HIP /
hipBLASLt / rocBLAS versions, gfx arch): on any other stack TunableOp ignores it and the default heuristic is used.
Measured: 761 -> 778 patches/s.
"""
import os

import torch

TUNED_FILE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "tunableop_gfx950.csv")


def enable_tuned_gemms(path=None):
    """Use the recorded GEMM solutions (no tuning at run time).  Returns True when the file was handed to TunableOp."""
    path = path or os.environ.get("DHZ_TUNABLEOP_FILE", TUNED_FILE)
    if os.environ.get("DHZ_NO_TUNED_GEMMS") or not os.path.exists(path) or not torch.cuda.is_available():
        return False
    tun = torch.cuda.tunable
    tun.enable(True)
    tun.tuning_enable(False)           # never tune inside a run: unknown shapes fall back to the default heuristic
    if hasattr(tun, "write_file_on_exit"):
        tun.write_file_on_exit(False)  # nothing new to record; N ranks must not race on a results file in the cwd
    try:
        return bool(tun.read_file(path))
    except Exception:                  # a stale / foreign file must never break a run
        return False
