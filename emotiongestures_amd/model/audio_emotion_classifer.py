"""Mirror of model/audio_emotion_classifer.py:17-49: EmotionNet, the audio emotion classifier, inference
forward on the HIP kernels: 4-stage ResNetSE ([3,4,6,3] SE blocks at 32/64/128/256 channels) on a [B,128,128] spectrogram
-> [B,256,16,16] -> 6-layer ReLU MLP -> 8 logits.  In train() mode the forward runs on the differentiable HIP operators of
emotiongestures_amd/train (batch-statistics BatchNorm); tools/train_emotion_net.py is the training loop of
train_audio_classifier_K_fold.py:109-200 on synthetic data."""
from __future__ import annotations

import torch
import torch.nn as nn

from .. import ops
from ..harness import _PackCache, _affine_chain
from ..modules import Linear, ReplicaAware, ResNetSE, SEBasicBlock, _eval_only, _seq, replica_forward


class EmotionNet(ReplicaAware, nn.Module):
    def __init__(self, *, precision="f32"):
        super().__init__()
        num_filters = [32, 64, 128, 256]
        self.emotion_encoder = ResNetSE(SEBasicBlock, [3, 4, 6, 3], num_filters)
        self.emotion_eocder_fc = _seq(Linear(256 * 16 * 16, 4096), None, Linear(4096, 2048), None, Linear(2048, 512), None,
                                      Linear(512, 128), None, Linear(128, 64), None)
        self.last_fc = Linear(64, 8)
        self.precision = precision
        self._cache = _PackCache()
        self._fc0 = None            # (weight version, packed image) of the first Linear with its columns in NHWC order

    def _first_fc_packed(self, device):
        """The feature map leaves the tower as NHWC; the reference flattens NCHW (feature.view(B, -1), :44).  Instead of
        transposing activations every call, the first Linear's 65536 columns are permuted (c,h,w) -> (h,w,c) once at pack time."""
        lin = self.emotion_eocder_fc[0]
        ver = (lin.weight._version, str(device))
        if self._fc0 is None or self._fc0[0] != ver:
            w = lin.weight.detach().view(-1, 256, 16, 16).permute(0, 2, 3, 1).reshape(lin.weight.shape[0], -1)
            self._fc0 = (ver, ops.pack_linear_weight(w, device))
        return self._fc0[1]

    @replica_forward
    def forward(self, mfcc):
        if self.training:           # train() mode (train_audio_classifier_K_fold.py:155-175): differentiable HIP operators
            from ..train import nets
            return nets.emotion_net_forward(self, mfcc)
        if mfcc.dim() != 3 or tuple(mfcc.shape[1:]) != (128, 128):
            raise ValueError(f"EmotionNet.forward: expected [B,128,128] (-> 256x16x16 features), got {tuple(mfcc.shape)}")
        for m in self.emotion_encoder.modules():
            if isinstance(m, SEBasicBlock):
                m.precision = self.precision
        feat = self.emotion_encoder.forward_nhwc(mfcc.contiguous())            # [B,16,16,256]
        B = feat.shape[0]
        lin0 = self.emotion_eocder_fc[0]
        # 65536-deep product at M = B rows: split K over 8 workgroup slices (64 column tiles x 8 = 512 workgroups), fixed-order reduce
        x = ops.linear_splitk(feat.reshape(B, -1), lin0.weight, lin0.bias, relu=True, splits=8, precision=self.precision,
                              packed=self._first_fc_packed(feat.device))
        # ReLU follows each of the five hidden layers (:25-35) and not last_fc (:37,46): one chain, ReLU between its layers
        rest = [self.emotion_eocder_fc[i] for i in (2, 4, 6, 8)] + [self.last_fc]
        return _affine_chain(self._cache, x, rest, True, self.precision)
