"""Drop-in for the reference's Uformer_ProbSparse/My_model.py: the dense-window-attention twin
(M0:428-518).  Same surface as My_model_1; WindowAttention has no `ProbSpare` sub-module, so the
state_dict matches the reference's dense checkpoints.  Every class exported here defaults to the dense
attention kernel (dhz_dense_attn_fwd/bwd)."""
from dehaze_hip import model as _m
from dehaze_hip.model import (Downsample, DropPath, InputProj, LeFF, LinearProjection, Mlp, OutputProj, Upsample,  # noqa: F401
                              to_2tuple, trunc_normal_, window_partition, window_reverse)
from dehaze_hip.unet import UNet  # noqa: F401


def _dense_default(cls):
    class _Dense(cls):
        def __init__(self, *args, **kwargs):
            kwargs.setdefault("variant", "dense")
            super().__init__(*args, **kwargs)
    _Dense.__name__ = _Dense.__qualname__ = cls.__name__
    _Dense.__doc__ = cls.__doc__
    return _Dense


WindowAttention = _dense_default(_m.WindowAttention)
LeWinTransformerBlock = _dense_default(_m.LeWinTransformerBlock)
BasicUformerLayer = _dense_default(_m.BasicUformerLayer)


class Uformer(_m.UformerDense):
    variant = "dense"
