"""Seeded synthetic batches with the reference's batch-dict layout {'input','target','names','A_emb'}
(trainUM.py:247-253; data/MedSpeckle.py:46,69-73).  The dataset itself is absent from the reference
snapshot (.MISSING_LARGE_BLOBS), so benchmarks and tests use these: smooth random GT
field in [-1,1], modality-specific degradations, L2-normalised image embedding."""
import torch
import torch.nn.functional as F

ARTIFACT_TYPES = ['speckle in OCT', 'speckle in ultra sound', 'noise in cryo-EM image', 'noise in low dose CT',
                  'Gaussian noise in MRI']  # Configurations/config.yml:15


def make_batch(B, H, W=None, seed=1234, mixed=True, M=1):
    W = W or H
    g = torch.Generator().manual_seed(seed)
    u = torch.rand(B, 1, H + 16, W + 16, generator=g)
    gt01 = F.avg_pool2d(F.avg_pool2d(u, 9, stride=1, padding=4), 9, stride=1, padding=4)[:, :, 8:8 + H, 8:8 + W]
    lo = gt01.amin(dim=(1, 2, 3), keepdim=True)
    hi = gt01.amax(dim=(1, 2, 3), keepdim=True)
    gt01 = (gt01 - lo) / (hi - lo).clamp_min(1e-6)
    names, lq = [], []
    for b in range(B):
        name = ARTIFACT_TYPES[b % 5] if mixed else ARTIFACT_TYPES[0]
        x = gt01[b]
        n = torch.randn(x.shape, generator=g)
        if "speckle" in name:
            y = x * (1 + 0.25 * n)
        elif "low dose CT" in name:
            y = x + 0.05 * torch.sqrt(x.clamp_min(0)) * n
        else:
            y = x + (25.0 / 255.0) * n  # config.yml:25, deg_utils.add_noise semantics
        names.append(name)
        lq.append(y)
    lq01 = torch.stack(lq)
    ge = torch.Generator().manual_seed(seed + 1)
    a_emb = F.normalize(torch.randn(B, M, 512, generator=ge), dim=-1)
    return {'input': (lq01 * 2 - 1).contiguous(), 'target': (gt01 * 2 - 1).contiguous(), 'names': names, 'A_emb': a_emb}
