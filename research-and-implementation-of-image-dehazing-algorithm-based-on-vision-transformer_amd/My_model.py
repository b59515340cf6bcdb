"""Drop-in for the reference's Uformer_ProbSparse/My_model.py: the dense-window-attention twin
(M0:428-518).  Same surface as My_model_1; WindowAttention has no `ProbSpare` sub-module, so the
state_dict matches the reference's dense checkpoints."""
from dehaze_hip.model import (BasicUformerLayer, Downsample, DropPath, InputProj, LeFF, LeWinTransformerBlock,  # noqa: F401
                              LinearProjection, OutputProj, Upsample, WindowAttention, to_2tuple, trunc_normal_,
                              window_partition, window_reverse)
from dehaze_hip.model import UformerDense as _UformerDense
from dehaze_hip.unet import UNet  # noqa: F401


class Uformer(_UformerDense):
    variant = "dense"
