"""Log-power STFT spectrogram on the GPU (replaces the per-item CPU librosa call in the
reference's DataLoader workers, core/dataset/dataset.py:461-495) + the audio window trim of
`_get_audio_segment` (:421-459, host integer arithmetic)."""
import numpy as np
import torch

from ..._lib import call, lib, ptr, stream_ptr, TbnHipError


def trim_audio_window(num_samples, frame_idx, audio_length, sampling_rate=24000, vid_fps=60):
    """(start, length) of the `audio_length` s window centred on frame_idx / fps, clamped like the reference
    (dataset.py:439-451): `sample = aud_sample[start : start + length]`.
    A clip SHORTER than the window (`num_samples < length`, dataset.py:441-446): the reference zero-pads the array to
    `length` samples but keeps clamping with the UN-updated `max_len`, so start = num_samples - length is NEGATIVE and the
    slice of the padded array runs from index `num_samples` to index `num_samples` -- an empty sample (which its librosa
    call then rejects).  The same (start, length) come back here; `trim_audio` applies them exactly as the reference does."""
    length = int(audio_length * sampling_rate)
    start_sec = float(frame_idx / vid_fps) - (audio_length / 2)
    start = int(max(0, start_sec * sampling_rate))
    if start + length > num_samples:
        start = num_samples - length
    return start, length


def trim_audio(aud_sample, frame_idx, audio_length, sampling_rate=24000, vid_fps=60):
    """reference `_get_audio_segment` (core/dataset/dataset.py:439-451) on a 1-D waveform -- NumPy array or torch tensor,
    host or device (a view, no copy, unless the clip must be padded): the `audio_length`-second window centred on
    frame_idx / fps, clamped to the clip.  A clip shorter than the window is zero-padded at its end (:441-442) and then
    sliced with the reference's own un-updated `max_len` (:447-450) -- Python's negative-start slice of the padded array,
    which is EMPTY; `Spectrogram` refuses an empty sample like the reference's librosa call does.  Nothing is "fixed"."""
    start, length = trim_audio_window(int(aud_sample.shape[0]), frame_idx, audio_length, sampling_rate, vid_fps)
    if aud_sample.shape[0] < length:
        pad = length - int(aud_sample.shape[0])
        if torch.is_tensor(aud_sample):
            aud_sample = torch.nn.functional.pad(aud_sample, (0, pad))
        else:
            aud_sample = np.pad(aud_sample, (0, pad))
    return aud_sample[start:start + length]


def _hz_to_mel(f):
    f = np.asarray(f, dtype=np.float64)
    f_sp, min_log_hz, logstep = 200.0 / 3, 1000.0, np.log(6.4) / 27.0
    return np.where(f >= min_log_hz, min_log_hz / f_sp + np.log(np.maximum(f, 1e-30) / min_log_hz) / logstep, f / f_sp)


def _mel_to_hz(m):
    m = np.asarray(m, dtype=np.float64)
    f_sp, min_log_hz, logstep = 200.0 / 3, 1000.0, np.log(6.4) / 27.0
    return np.where(m >= min_log_hz / f_sp, min_log_hz * np.exp(logstep * (m - min_log_hz / f_sp)), f_sp * m)


def mel_filterbank(sr=24000, n_fft=511, n_mels=128):
    """librosa.filters.mel(sr, n_fft, n_mels) (Slaney scale and normalisation): (n_mels, 256) float32"""
    fmax = sr / 2.0
    freqs = np.linspace(0, fmax, 1 + n_fft // 2, endpoint=True)
    mel_f = _mel_to_hz(np.linspace(_hz_to_mel(0.0), _hz_to_mel(fmax), n_mels + 2))
    fdiff = np.diff(mel_f)
    ramps = mel_f[:, None] - freqs[None, :]
    w = np.maximum(0, np.minimum(-ramps[:-2] / fdiff[:-1, None], ramps[2:] / fdiff[1:, None]))
    return (w * (2.0 / (mel_f[2:] - mel_f[:-2]))[:, None]).astype(np.float32)


class Spectrogram:
    """`spec = Spectrogram()(wave)`: wave (nseg, L) float32 on the GPU -> (nseg, 256, 1+(L-1)//120) log-power STFT
    (`spec_type="stft"`, the reference default) or (nseg, 128, T) log-mel dB (`spec_type="logms"`,
    dataset.py:496-506: librosa melspectrogram + power_to_db(ref=max) of each segment)."""

    def __init__(self, eps=1e-6, spec_type="stft", sampling_rate=24000):
        if spec_type not in ("stft", "logms"):
            raise Exception("Unknown spectrogram representation")
        self.eps = eps
        self.spec_type = spec_type
        self.sampling_rate = sampling_rate
        self._tw = {}
        self._mel = {}

    def _melbasis(self, device):
        if device not in self._mel:
            self._mel[device] = torch.from_numpy(mel_filterbank(self.sampling_rate)).to(device)
        return self._mel[device]

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
        if L < 1:       # librosa 0.7.2 (reference dataset.py:487-495) rejects it too: "Input is too short" (util.frame)
            raise ValueError("Spectrogram: empty audio sample (a clip shorter than audio_length, reference dataset.py:441-451)")
        W = 1 + (L - 1) // 120
        spec = torch.empty(nseg, 256, W, device=wave.device, dtype=torch.float32)
        eps = float(self.eps) if self.spec_type == "stft" else 0.0
        call("tbn_stft_logpower", ptr(wave), nseg, L, ptr(self._twiddle(wave.device)), ptr(spec), eps, stream_ptr())
        if self.spec_type == "stft":
            return spec
        # log-mel: the 128 x 256 mel projection of the power spectrum is a (tiny) HIP GEMM, the dB conversion is
        # relative to each segment's own maximum (librosa.power_to_db(S, ref=np.max), top_db = 80)
        from ... import ops
        power = torch.exp(spec)                                   # kernel returns log(power + 0)
        rows = power.permute(0, 2, 1).reshape(nseg * W, 256)       # (segment, frame) rows x frequency
        mel = ops.linear(rows, self._melbasis(wave.device), None).reshape(nseg, W, 128).permute(0, 2, 1)
        amin = 1e-10
        ref = mel.amax(dim=(1, 2), keepdim=True)
        db = 10.0 * torch.log10(torch.clamp(mel, min=amin)) - 10.0 * torch.log10(torch.clamp(ref, min=amin))
        return torch.maximum(db, db.amax(dim=(1, 2), keepdim=True) - 80.0).contiguous()
