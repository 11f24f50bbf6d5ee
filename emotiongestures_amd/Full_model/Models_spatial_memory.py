"""Mirror of Full_model/Models_spatial_memory.py: Transformer (:471-616) with SP_Memory_Net_v2 (:255-295)."""
from ..modules import (Audio_ResNetEncoder, Decoder, Encoder, PositionalEncoding, Prior_MemoryEncoder,  # noqa: F401
                       SP_Memory_Net_v1, SP_Memory_Net_v2, TM_Memory_Net, TextEncoderTCN, Transformer)
from ..harness import Motion_Discriminator, Pose_Discriminator  # noqa: E402,F401  (:620-669, :671-704 upstream)
