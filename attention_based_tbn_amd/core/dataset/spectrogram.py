"""Log-power STFT spectrogram on the GPU (replaces the per-item CPU librosa call in the
reference's DataLoader workers, core/dataset/dataset.py:461-495) + the audio window trim of
`_get_audio_segment` (:421-459, host integer arithmetic)."""
import numpy as np
import torch

from ..._lib import call, lib, ptr, stream_ptr, TbnHipError


def trim_audio_window(num_samples, frame_idx, audio_length, sampling_rate=24000, vid_fps=60):
    """(start, length) of the `audio_length` s window centred on frame_idx / fps, clamped like the
    reference (dataset.py:439-451).  The clip is assumed to be at least `length` samples long."""
    length = int(audio_length * sampling_rate)
    start_sec = float(frame_idx / vid_fps) - (audio_length / 2)
    start = int(max(0, start_sec * sampling_rate))
    if start + length > num_samples:
        start = num_samples - length
    return start, length


class Spectrogram:
    """`spec = Spectrogram()(wave)`: wave (nseg, L) float32 on the GPU -> (nseg, 256, 1+(L-1)//120)"""

    def __init__(self, eps=1e-6):
        self.eps = eps
        self._tw = {}

    def _twiddle(self, device):
        if device not in self._tw:
            n = lib().tbn_stft_twiddle_floats()
            host = np.empty(n, dtype=np.float32)
            call("tbn_stft_make_twiddle", host.ctypes.data)
            self._tw[device] = torch.from_numpy(host).to(device)
        return self._tw[device]

    def __call__(self, wave):
        if not wave.is_cuda:
            raise TbnHipError("Spectrogram: the STFT kernel needs the waveform on the GPU (no CPU fallback)")
        wave = wave.contiguous().float()
        nseg, L = wave.shape
        W = 1 + (L - 1) // 120
        spec = torch.empty(nseg, 256, W, device=wave.device, dtype=torch.float32)
        call("tbn_stft_logpower", ptr(wave), nseg, L, ptr(self._twiddle(wave.device)), ptr(spec), float(self.eps),
             stream_ptr())
        return spec
