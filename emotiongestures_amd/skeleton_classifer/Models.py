"""Mirror of skeleton_classifer/Models.py: Transformer (:199-283), Prior_Encoder (:88-116) on the HIP path."""
from ..harness import Prior_Encoder  # noqa: F401
from ..harness import SkeletonTransformer as Transformer  # noqa: F401
from ..modules import Encoder  # noqa: F401
