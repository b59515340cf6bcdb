"""Drop-in for Uformer_ProbSparse/My_CR.py: Vgg19 feature slicer (My_CR.py:56-86) and ContrastLoss
(My_CR.py:89-123), device-agnostic (the reference hard-codes `.cuda()`, My_CR.py:94).

VGG19 weights: the reference pulls torchvision's ImageNet checkpoint at run time (My_CR.py:59).  That
file is a third-party artefact and there is no network here, so `Vgg19` loads a user-supplied
state_dict when given (`weights=` or $DEHAZE_VGG19_WEIGHTS, torchvision `features.N.*` or
`sliceK.N.*` keys) and otherwise falls back to SEEDED RANDOM weights (same FLOPs; flagged with a
warning).  No ImageNet mean/std normalisation is applied - the reference applies none either.
"""
import os
import warnings

import torch
import torch.nn as nn
import torch.nn.functional as F

# torchvision cfg 'E' up to features[29]; slice boundaries of My_CR.py:65-74
_CFG = (64, 64, 'M', 128, 128, 'M', 256, 256, 256, 256, 'M', 512, 512, 512, 512, 'M', 512)
_SLICES = ((0, 2), (2, 7), (7, 12), (12, 21), (21, 30))


def _feature_layers():
    layers, cin = [], 3
    for v in _CFG:
        if v == 'M':
            layers.append(nn.MaxPool2d(kernel_size=2, stride=2))
        else:
            layers += [nn.Conv2d(cin, v, kernel_size=3, padding=1), nn.ReLU(inplace=True)]
            cin = v
    return layers          # 30 entries = torchvision vgg19().features[0:30]


def seeded_vgg_init_(layers, seed=1905):
    g = torch.Generator().manual_seed(seed)
    full = (64, 64, 'M', 128, 128, 'M', 256, 256, 256, 256, 'M', 512, 512, 512, 512, 'M', 512, 512, 512, 512, 'M')
    convs = [m for m in layers if isinstance(m, nn.Conv2d)]
    ci, cin = 0, 3
    for v in full:                      # draw for all 16 convs so the stream matches a full vgg19 init
        if v == 'M':
            continue
        w = torch.randn(v, cin, 3, 3, generator=g) * (2.0 / (cin * 9)) ** 0.5
        b = torch.randn(v, generator=g) * 0.05
        if ci < len(convs):
            convs[ci].weight.data.copy_(w)
            convs[ci].bias.data.copy_(b)
        ci += 1
        cin = v


class Vgg19(nn.Module):
    def __init__(self, requires_grad=False, weights=None):
        super().__init__()
        feats = _feature_layers()
        for k, (lo, hi) in enumerate(_SLICES, 1):
            seq = nn.Sequential()
            for x in range(lo, hi):
                seq.add_module(str(x), feats[x])
            setattr(self, f"slice{k}", seq)
        weights = weights or os.environ.get("DEHAZE_VGG19_WEIGHTS")
        if weights:
            sd = torch.load(weights, map_location="cpu")
            sd = sd.get("state_dict", sd)
            mine = self.state_dict()
            for k in mine:
                n = k.split(".", 1)[1]                      # 'slice3.7.weight' -> '7.weight'
                src = sd.get(k, sd.get("features." + n))
                if src is None:
                    raise KeyError(f"VGG19 weight file lacks {k} / features.{n}")
                mine[k].copy_(src)
            self.pretrained = True
        else:
            warnings.warn("Vgg19: no ImageNet checkpoint available offline - using SEEDED RANDOM weights "
                          "(set DEHAZE_VGG19_WEIGHTS=/path/to/vgg19-dcbb9e9d.pth for the reference behaviour)")
            seeded_vgg_init_(feats)
            self.pretrained = False
        if not requires_grad:
            for p in self.parameters():
                p.requires_grad = False
        self._frozen = not requires_grad
        self._engine = None
        self.feature_dtype = torch.float32

    def engine_for(self, X):
        """The MFMA feature engine (dehaze_hip/vgg.py) when it applies: frozen filters, an fp32 image on the GPU, maps that stay
        multiples of 16 down to relu4_4 (H, W multiples of 128); None -> library convolutions.  `feature_dtype` selects the
        Winograd fp32 engine (default) or, for BASELINE config 4, the implicit-GEMM engine with bf16 feature maps - what
        torch.autocast makes of these convolutions in the reference."""
        if not (self._frozen and X.is_cuda and X.dtype == torch.float32 and X.dim() == 4 and X.shape[1] == 3
                and X.shape[2] % 128 == 0 and X.shape[3] % 128 == 0 and self.slice1[0].weight.device == X.device):
            return None                      # (filters on another device: let the library path raise torch's own error)
        bf16 = self.feature_dtype == torch.bfloat16
        if self._engine is None or self._engine[0] != bf16:
            from dehaze_hip.vgg import VggEngine, VggEngineBF16
            self._engine = (bf16, (VggEngineBF16 if bf16 else VggEngine)([m for m in self.modules() if isinstance(m, nn.Conv2d)]))
        return self._engine[1]

    def forward(self, X):
        eng = self.engine_for(X)
        if eng is not None:
            from dehaze_hip import vgg as _v
            taps = _v.vgg_taps(eng, X)
            if self.feature_dtype == torch.bfloat16:
                return [t.permute(0, 3, 1, 2) for t in taps]          # NHWC bf16 maps seen as NCHW (autocast's output dtype)
            return [_v.to_plain_tap(t) for t in taps]
        h1 = self.slice1(X)
        h2 = self.slice2(h1)
        h3 = self.slice3(h2)
        h4 = self.slice4(h3)
        h5 = self.slice5(h4)
        return [h1, h2, h3, h4, h5]


class ContrastLoss(nn.Module):
    """sum_i w_i * L1(a_i,p_i) / (L1(a_i,n_i) + 1e-7), w = [1/32,1/16,1/8,1/4,1]; ablation -> numerator
    only.  Returns (loss, all_ap, all_an) like the reference."""

    def __init__(self, ablation=False, weights=None):
        super().__init__()
        self.vgg = Vgg19(weights=weights)
        self.l1 = nn.L1Loss()
        self.weights = [1.0 / 32, 1.0 / 16, 1.0 / 8, 1.0 / 4, 1.0]
        self.ab = ablation

    def reference_taps(self, p, n):
        """The no-gradient half of forward(): the feature taps of (p, n) in the engine's layout, or None when the engine does not
        apply.  train_step can run it on a side stream beside the model's forward (the inputs exist before the step starts) and
        hand the result to forward(..., pn_taps=)."""
        eng = self.vgg.engine_for(p)
        if eng is None:
            return None
        from dehaze_hip.vgg import vgg_taps
        with torch.no_grad():
            return vgg_taps(eng, torch.cat([p, n], 0) if not self.ab else p)

    def forward(self, a, p, n, pn_taps=None):
        # p and n never need gradients: one batched VGG pass for both, one (with grad) for a
        eng = self.vgg.engine_for(a)
        if eng is not None:
            # features stay in the engine's channel-blocked layout: the L1 means do not care about element order
            from dehaze_hip.vgg import vgg_taps
            pn = pn_taps if pn_taps is not None else self.reference_taps(p, n)
            a_vgg = vgg_taps(eng, a)
        else:
            with torch.no_grad():
                pn = self.vgg(torch.cat([p, n], 0)) if not self.ab else self.vgg(p)
            a_vgg = self.vgg(a)
        B = a.shape[0]
        if eng is not None and all(t.numel() % 4 == 0 for t in a_vgg):
            # every tap: both distances in one pass over (a, p, n); the scalar combination of all taps is one autograd node
            from dehaze_hip.vgg import contrast_taps
            return contrast_taps(a_vgg, pn, B, self.weights, self.ab)
        loss, all_ap, all_an = 0, 0, 0
        for i in range(len(a_vgg)):
            if eng is not None and a_vgg[i].numel() % 4 == 0:
                # both distances of this tap in one pass over (a, p, n), joint backward in another
                from dehaze_hip.vgg import l1_pair
                d = l1_pair(a_vgg[i], pn[i][:B], None if self.ab else pn[i][B:])
                d_ap = d[0]
                all_ap = all_ap + d_ap
                if not self.ab:
                    all_an = all_an + d[1]
                    loss = loss + self.weights[i] * (d_ap / (d[1] + 1e-7))
                else:
                    loss = loss + self.weights[i] * d_ap
                continue
            d_ap = self.l1(a_vgg[i], pn[i][:B])
            all_ap = all_ap + d_ap
            if not self.ab:
                d_an = self.l1(a_vgg[i], pn[i][B:])
                all_an = all_an + d_an
                contrastive = d_ap / (d_an + 1e-7)
            else:
                contrastive = d_ap
            loss = loss + self.weights[i] * contrastive
        return loss, all_ap, all_an
