"""Training path of the EmotionGesture hot path on the HIP kernels.

functional.py  differentiable operators (forward and backward are HIP kernels; fp32)
nets.py        train-mode forwards of EmotionNet and the generator (batch-statistics BatchNorm)
optim.py       flat parameter / gradient buffers, fused Adam, bucketed gradient all-reduce (RCCL) for data parallelism
"""
from . import functional, nets, optim  # noqa: F401
from .nets import emotion_net_forward, generator_forward  # noqa: F401
from .optim import FlatAdam, GradBuckets, flatten_parameters  # noqa: F401
