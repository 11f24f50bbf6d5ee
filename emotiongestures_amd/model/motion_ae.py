"""Mirror of model/motion_ae.py: MotionAE (:118-130), the 34-frame pose auto-encoder behind the TED-style Frechet gesture
distance (model/embedding_space_evaluator.py:27-31), eval-mode forward on the HIP kernels.

Every BatchNorm1d directly follows a Conv1d / Linear, so it is folded into that layer's weights when the layer is packed.
`nn.LeakyReLU(True)` in the dense stacks (:48,51,84,92) has negative_slope == True == 1.0, i.e. it is the identity, which turns
out_net / pre_net into pure affine chains.  ConvTranspose1d(k=3, stride 1, no padding) equals Conv1d with the kernel
flipped, in/out channels swapped and padding 2."""
from __future__ import annotations

import torch
import torch.nn as nn

from .. import ops
from ..modules import BatchNorm, Conv1d, ConvTranspose1d, Linear, ReplicaAware, _eval_only, _seq, replica_forward


def ConvNormRelu(in_channels, out_channels, downsample=False, padding=0, batchnorm=True):
    """model/motion_ae.py:8-31"""
    k, s = (4, 2) if downsample else (3, 1)
    conv = Conv1d(in_channels, out_channels, kernel_size=k, stride=s, padding=padding)
    return _seq(conv, BatchNorm(out_channels), None) if batchnorm else _seq(conv, None)


def _fold(weight, bias, bn: BatchNorm):
    s, t = bn.affine()
    shape = (-1,) + (1,) * (weight.dim() - 1)
    return weight * s.view(shape), bias * s + t


def _versions(*mods):
    return tuple(t._version for m in mods if m is not None for t in list(m.parameters()) + list(m.buffers()))


def _dense(x, lin: Linear, bn=None):
    """Linear (+ folded BatchNorm1d); the folded, packed weight image is cached on the layer per parameter version."""
    ver = (str(x.device),) + _versions(lin, bn)
    hit = getattr(lin, "_eg_pack", None)
    if hit is None or hit[0] != ver:
        w, b = (lin.weight, lin.bias) if bn is None else _fold(lin.weight, lin.bias, bn)
        hit = (ver, w.detach(), b.detach().to(x.device).contiguous(), ops.pack_linear_weight(w, x.device))
        object.__setattr__(lin, "_eg_pack", hit)
    return ops.linear(x, hit[1], hit[2], packed=hit[3])


def _conv(x, conv, bn=None, weight=None, stride=1, padding=0, leaky=False):
    """Conv1d (+ folded BatchNorm1d, + LeakyReLU 0.2); folded weights cached on the layer per parameter version.
    ``weight`` overrides conv.weight (ConvTranspose1d expressed as a convolution)."""
    ver = (str(x.device),) + _versions(conv, bn)
    hit = getattr(conv, "_eg_pack", None)
    if hit is None or hit[0] != ver:
        w = conv.weight if weight is None else weight(conv)
        w, b = (w, conv.bias) if bn is None else _fold(w, conv.bias, bn)
        hit = (ver, w.detach().float().contiguous().to(x.device), b.detach().float().contiguous().to(x.device))
        object.__setattr__(conv, "_eg_pack", hit)
    return ops.conv1d(x, hit[1], hit[2], stride=stride, padding=padding, leaky=leaky)


class PoseEncoderConv(nn.Module):
    """model/motion_ae.py:33-63"""

    def __init__(self, length, pose_dim, latent_dim):
        super().__init__()
        if length != 34:
            raise ValueError("PoseEncoderConv: out_net is sized for 34 frames upstream (Linear(384, 256), :46)")
        self.net = _seq(ConvNormRelu(pose_dim, 32), ConvNormRelu(32, 64), ConvNormRelu(64, 64, True), Conv1d(64, 32, 3))
        self.out_net = _seq(Linear(384, 256), BatchNorm(256), None, Linear(256, 128), BatchNorm(128), None, Linear(128, latent_dim))

    def forward(self, poses):
        _eval_only(self)
        x = poses.transpose(1, 2).contiguous()                      # [B, dim, seq]
        for blk in (self.net[0], self.net[1], self.net[2]):
            x = _conv(x, blk[0], blk[1], stride=blk[0].stride, padding=blk[0].padding, leaky=True)
        x = _conv(x, self.net[3])
        x = x.flatten(1).contiguous()
        x = _dense(x, self.out_net[0], self.out_net[1])
        x = _dense(x, self.out_net[3], self.out_net[4])
        return _dense(x, self.out_net[6])


class PoseDecoderConv(nn.Module):
    """model/motion_ae.py:65-116 (use_pre_poses=False as built by MotionAE)."""

    def __init__(self, length, pose_dim, latent_dim, use_pre_poses=False):
        super().__init__()
        if use_pre_poses or length != 34:
            raise ValueError("PoseDecoderConv: only the MotionAE configuration (34 frames, no pre-poses) is built")
        self.use_pre_poses = False
        self.pre_net = _seq(Linear(latent_dim, 64), BatchNorm(64), None, Linear(64, 136))
        self.net = _seq(ConvTranspose1d(4, 32, 3, stride=1, padding=0, output_padding=0), BatchNorm(32), None,
                        ConvTranspose1d(32, 32, 3, stride=1, padding=0, output_padding=0), BatchNorm(32), None,
                        Conv1d(32, 32, 3), Conv1d(32, pose_dim, 3))

    @staticmethod
    def _as_conv(ct: ConvTranspose1d):
        return ct.weight.permute(1, 0, 2).flip(2)                   # [cin, cout, k] -> conv weight [cout, cin, k], taps reversed

    def forward(self, feat, pre_poses=None):
        _eval_only(self)
        x = _dense(feat.contiguous(), self.pre_net[0], self.pre_net[1])
        x = _dense(x, self.pre_net[3]).view(feat.shape[0], 4, -1).contiguous()
        for ct, bn in ((self.net[0], self.net[1]), (self.net[3], self.net[4])):
            x = _conv(x, ct, bn, weight=self._as_conv, padding=2, leaky=True)
        x = _conv(x, self.net[6])
        x = _conv(x, self.net[7])
        return x.transpose(1, 2)


class MotionAE(ReplicaAware, nn.Module):
    def __init__(self, pose_dim, latent_dim):
        super().__init__()
        self.encoder = PoseEncoderConv(34, pose_dim, latent_dim)
        self.decoder = PoseDecoderConv(34, pose_dim, latent_dim)

    @replica_forward
    def forward(self, pose):
        pose = pose.view(pose.size(0), pose.size(1), -1)
        z = self.encoder(pose)
        return self.decoder(z), z
