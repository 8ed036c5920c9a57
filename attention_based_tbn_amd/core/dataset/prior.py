"""Ground-truth attention priors of the reference data layer (`core/dataset/dataset.py:534-575`,
`Video_Dataset._get_attn_weights`): the target the `prior` loss (`core/models/model.py:291-311`) pulls the audio
attention towards.  Host NumPy (a (T, 1) vector per segment, T = round(audio_length * 25 / 4) <= 25).

  gaussian  cv2.getGaussianKernel(T, sigma=1)        -- restated: exp(-(i - (T-1)/2)^2 / 2) / sum   (float64)
  uniform   ones / T                                  (float32)
  loud      the Gaussian rolled onto the loudest window of the spectrogram, floor outside +-4 bins
"""
import numpy as np
import torch


def gaussian_kernel(n, sigma=1.0):
    """cv2.getGaussianKernel(n, sigma) for sigma > 0: (n, 1) float64, normalised"""
    x = np.arange(n, dtype=np.float64) - (n - 1) * 0.5
    k = np.exp(-(x * x) / (2.0 * sigma * sigma))
    return (k / k.sum()).reshape(n, 1)


def attention_prior(spec, audio_length, prior_type):
    """spec: (256, W) log-spectrogram of the segment (only read by 'loud') -> float32 tensor (T, 1)"""
    anchor = 25 / 4
    win_size = round(audio_length * anchor)
    if prior_type == "gaussian":
        wts = gaussian_kernel(win_size, 1)
    elif prior_type == "uniform":
        wts = np.ones((win_size, 1), dtype=np.float32) / win_size
    elif prior_type == "loud":
        spec = np.asarray(spec)
        loudness = np.array([np.max(spec[:, i:i + win_size]) for i in range(0, spec.shape[1], win_size)
                             if i + win_size <= spec.shape[1]])
        loudest_loc = loudness.argsort()[-1]
        wts = gaussian_kernel(win_size, 1)
        min_val = wts.min()
        mean_loc = wts.shape[0] // 2
        new_mean_loc = loudest_loc
        if new_mean_loc <= wts.shape[0] and (new_mean_loc < mean_loc - 2 or new_mean_loc > mean_loc + 2):
            wts = np.roll(wts, new_mean_loc - mean_loc)
            if new_mean_loc - 4 > 0:
                wts[:new_mean_loc - 4] = min_val
            if new_mean_loc + 4 < wts.shape[0]:
                wts[new_mean_loc + 4:] = min_val
        wts = np.stack([wts]).mean(0)
    else:
        raise ValueError(f"unknown attention prior '{prior_type}'")
    return torch.tensor(wts).float()
