"""Drop-in for Uformer_ProbSparse/ProbSparse/attn.py: AttentionLayer (ATT:345-461) and ProbAttention
(ATT:43-342) backed by the fused HIP kernel `dhz_ps_attn_fwd/bwd`."""
from dehaze_hip.model import AttentionLayer, ProbAttention  # noqa: F401
