"""ORACLE (test infrastructure): RMSE / PSNR / SSIM exactly as the reference drivers compute them
(trainUM.py:319-329, testUM.py:151-164) with skimage -- which is not installed here, so skimage 0.19+'s published
`structural_similarity` algorithm is restated with scipy.ndimage.gaussian_filter (the function skimage itself
calls): gaussian_weights=True, sigma=1.5 -> truncate 3.5 (radius 5, 11x11), use_sample_covariance=False
(cov_norm = 1), K1=.01, K2=.03, data_range=1, mean over the image cropped by (win_size-1)//2.  "parity
unpinned" against skimage itself (absent); pinned only by this restatement."""
import numpy as np
from scipy.ndimage import gaussian_filter


def metrics(pred, target):
    """pred/target: 2-D arrays in [-1,1]; returns (RMSE, PSNR, SSIM) on x/2+0.5 in float64"""
    p = np.asarray(pred, dtype=np.float64) / 2 + 0.5
    g = np.asarray(target, dtype=np.float64) / 2 + 0.5
    mse = np.mean((p - g) ** 2)
    rmse = np.sqrt(mse)
    psnr = 10 * np.log10(1.0 / mse)
    kw = dict(sigma=1.5, truncate=3.5, mode="reflect")
    ux, uy = gaussian_filter(p, **kw), gaussian_filter(g, **kw)
    uxx, uyy, uxy = gaussian_filter(p * p, **kw), gaussian_filter(g * g, **kw), gaussian_filter(p * g, **kw)
    vx, vy, vxy = uxx - ux * ux, uyy - uy * uy, uxy - ux * uy
    C1, C2 = 0.01 ** 2, 0.03 ** 2
    S = ((2 * ux * uy + C1) * (2 * vxy + C2)) / ((ux ** 2 + uy ** 2 + C1) * (vx + vy + C2))
    pad = 5
    return rmse, psnr, float(S[pad:-pad, pad:-pad].mean())
