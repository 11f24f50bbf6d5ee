"""Mirror of Full_model/Models_memory.py: Transformer (:426-565) with SP_Memory_Net_v1 + TM_Memory_Net (:215-293).
This is the variant the reference's eval script imports (test_emotion_gesture_diversity_iterative.py:25)."""
from ..modules import (Audio_ResNetEncoder, Decoder, Encoder, PositionalEncoding, Prior_MemoryEncoder,  # noqa: F401
                       SP_Memory_Net_v1, TM_Memory_Net, TextEncoderTCN)
from ..modules import TransformerMemory as Transformer  # noqa: F401
from ..harness import Motion_Discriminator  # noqa: E402,F401  (:569-618 upstream; imported by the eval script beside Transformer)
