"""Mirror of model/FHD_score.py: calculate_frechet_distance (:159-217), diversity_score / calculate_diversity (:247-311).
``diversity_score(activations, device)`` keeps the upstream signature; ``device`` is ignored (host-side float64 as upstream
does after its .cpu())."""
from ..harness import calculate_diversity, calculate_frechet_distance  # noqa: F401
from ..harness import diversity_score as _diversity_score


def diversity_score(activations, device=None, frames=60):
    return _diversity_score(activations, frames)
