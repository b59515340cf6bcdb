"""MixUp (utils/dataset_utils.py:43-63): host randperm + Beta(1.2,1.2) draws, device-agnostic."""
import torch


class MixUp_AUG:
    def __init__(self):
        self.dist = torch.distributions.beta.Beta(torch.tensor([1.2]), torch.tensor([1.2]))

    def aug(self, rgb_gt, rgb_noisy):
        bs = rgb_gt.size(0)
        indices = torch.randperm(bs)
        idx_dev = indices.to(rgb_gt.device)
        rgb_gt2, rgb_noisy2 = rgb_gt[idx_dev], rgb_noisy[idx_dev]
        lam = self.dist.rsample((bs, 1)).view(-1, 1, 1, 1).to(rgb_gt.device)
        return lam * rgb_gt + (1 - lam) * rgb_gt2, lam * rgb_noisy + (1 - lam) * rgb_noisy2
