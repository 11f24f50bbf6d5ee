"""Mirror of CAVE/BEAT_CVAE.py: MLP_Reconstruct_v3 (:312-460) on the HIP path."""
from ..modules import MLP_Reconstruct_v3  # noqa: F401
