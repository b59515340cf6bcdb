"""Drop-in for Uformer_ProbSparse/losses.py: CharbonnierLoss (losses.py:41-52) on the fused HIP
reduction kernel (dhz_charbonnier_fwd/bwd)."""
import torch.nn as nn

from dehaze_hip import ops


class CharbonnierLoss(nn.Module):
    """mean(sqrt((x-y)^2 + eps^2)), eps = 1e-3."""

    def __init__(self, eps=1e-3):
        super().__init__()
        self.eps = eps

    def forward(self, x, y):
        return ops.charbonnier(x, y, self.eps)

    def forward_clamped(self, x, y):
        """(loss(clamp(x,0,1), y), clamp(x,0,1)) in one pass - the train step's TR:230+234."""
        return ops.charbonnier_clamped(x, y, self.eps)
