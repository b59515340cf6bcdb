"""Drop-in for Uformer_ProbSparse/utils/dataset_utils.py: the 8 rotate/flip augmentations (:6-40) and MixUp (:43-63;
host randperm + Beta(1.2,1.2) draws, device-agnostic - the reference hard-codes .cuda())."""
import torch


class Augment_RGB_torch:
    def __init__(self):
        pass

    def transform0(self, t):
        return t

    def transform1(self, t):
        return torch.rot90(t, k=1, dims=[-1, -2])

    def transform2(self, t):
        return torch.rot90(t, k=2, dims=[-1, -2])

    def transform3(self, t):
        return torch.rot90(t, k=3, dims=[-1, -2])

    def transform4(self, t):
        return t.flip(-2)

    def transform5(self, t):
        return torch.rot90(t, k=1, dims=[-1, -2]).flip(-2)

    def transform6(self, t):
        return torch.rot90(t, k=2, dims=[-1, -2]).flip(-2)

    def transform7(self, t):
        return torch.rot90(t, k=3, dims=[-1, -2]).flip(-2)


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
