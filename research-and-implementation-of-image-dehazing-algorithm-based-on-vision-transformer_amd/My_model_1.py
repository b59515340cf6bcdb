"""Drop-in for the reference's Uformer_ProbSparse/My_model_1.py (the model `utils.get_arch` builds,
model_utils.py:81): ProbSparse window attention Uformer on the MI355X HIP kernels.  Same constructor
signature (M1:961-967), module tree, parameter order and state_dict keys as the reference."""
from dehaze_hip.model import (AttentionLayer, BasicUformerLayer, Downsample, DropPath, InputProj, LeFF, Mlp,  # noqa: F401
                              ConvProjection, LeWinTransformerBlock, LinearProjection, LinearProjection_Concat_kv, OutputProj, ProbAttention,
                              SELayer, SepConv2d, Upsample,
                              WindowAttention, to_2tuple, trunc_normal_, window_partition, window_reverse)
from dehaze_hip.model import Uformer as _Uformer
from dehaze_hip.unet import UNet  # noqa: F401


class Uformer(_Uformer):
    variant = "probsparse"
