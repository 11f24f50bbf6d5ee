"""Mirror of model/embedding_space_evaluator.py:16-154 for the MotionAE branch (pose_dim 126): latent features of real and
generated 34-frame clips -> Frechet distance + mean L1 feature distance; reconstruction / cosine error differences.
The EmbeddingNet branch (pose_dim 27, a different dataset's model) is not built; `get_features_for_viz` needs `umap`."""
from __future__ import annotations

import numpy as np
import torch
import torch.nn.functional as F

from .motion_ae import MotionAE


class EmbeddingSpaceEvaluator:
    def __init__(self, args, embed_net_path, lang_model, device):
        self.n_pre_poses = args.n_pre_poses
        ckpt = embed_net_path if isinstance(embed_net_path, dict) else torch.load(embed_net_path, map_location="cpu")
        self.pose_dim = ckpt["pose_dim"]
        if args.pose_dim != 126 or "motion_ae" not in ckpt:
            raise NotImplementedError("EmbeddingSpaceEvaluator: only the MotionAE branch (args.pose_dim == 126) is built")
        self.latent_dim = ckpt["latent_dim"]
        self.net = MotionAE(self.pose_dim, self.latent_dim)
        self.net.load_state_dict(ckpt["motion_ae"])
        self.net.to(device).train(False)
        self.reset()

    def reset(self):
        self.context_feat_list, self.real_feat_list, self.generated_feat_list = [], [], []
        self.recon_err_diff, self.cos_err_diff = [], []

    def get_no_of_samples(self):
        return len(self.real_feat_list)

    @staticmethod
    def _errors(recon, poses):
        """:78-88: per-clip mean L1 of poses and of frame differences, summed over the batch; summed (1 - cosine) per 3-vector."""
        loss = torch.mean(F.l1_loss(recon, poses, reduction="none"), dim=(1, 2))
        loss = loss + torch.mean(F.l1_loss(recon[:, 1:] - recon[:, :-1], poses[:, 1:] - poses[:, :-1], reduction="none"), dim=(1, 2))
        b, t = poses.shape[:2]
        cos = torch.sum(1 - torch.cosine_similarity(recon.reshape(b, t, -1, 3), poses.reshape(b, t, -1, 3), dim=-1))
        return torch.sum(loss), cos

    @torch.no_grad()
    def push_samples(self, context_text, context_spec, generated_poses, real_poses):
        real_recon, real_feat = self.net(real_poses)
        generated_recon, generated_feat = self.net(generated_poses)
        self.real_feat_list.append(real_feat.cpu().numpy())
        self.generated_feat_list.append(generated_feat.cpu().numpy())
        l_real, c_real = self._errors(real_recon, real_poses.view(real_poses.size(0), real_poses.size(1), -1))
        l_fake, c_fake = self._errors(generated_recon, generated_poses.view(generated_poses.size(0), generated_poses.size(1), -1))
        self.recon_err_diff.append(l_fake - l_real)
        self.cos_err_diff.append(c_fake - c_real)

    def get_features_for_viz(self):
        import umap
        g, r = np.vstack(self.generated_feat_list), np.vstack(self.real_feat_list)
        t = umap.UMAP().fit_transform(np.vstack((g, r)))
        n = int(t.shape[0] / 2)
        return t[n:, :], t[0:n, :]

    def get_diversity_scores(self):
        feat1 = np.vstack(self.generated_feat_list[:500])
        idx = torch.randperm(len(self.generated_feat_list))[:500]
        feat2 = np.vstack([self.generated_feat_list[i] for i in idx])
        return np.mean(np.sum(np.absolute(feat1 - feat2), axis=-1))

    def get_scores(self):
        g, r = np.vstack(self.generated_feat_list), np.vstack(self.real_feat_list)
        try:
            fd = self.calculate_frechet_distance(np.mean(g, axis=0), np.cov(g, rowvar=False), np.mean(r, axis=0), np.cov(r, rowvar=False))
        except ValueError:
            fd = 1e+10
        feat_dist = np.mean(np.sum(np.absolute(r - g), axis=-1))
        return fd, feat_dist

    @staticmethod
    def calculate_frechet_distance(mu1, sigma1, mu2, sigma2, eps=1e-6):
        """:156-209: the Frechet distance of harness.calculate_frechet_distance; unlike model/FHD_score.py's variant a non-negligible
        imaginary part raises ValueError (get_scores maps that to 1e10)."""
        from ..harness import calculate_frechet_distance as frechet
        return frechet(mu1, sigma1, mu2, sigma2, eps=eps, imaginary="raise")
