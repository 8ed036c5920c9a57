"""Oracle (TEST INFRASTRUCTURE): log-power STFT spectrogram + audio window trim, NumPy.

Restates reference `core/dataset/dataset.py:461-495` (`_get_spectrogram`, spec_type
"stft") and `:421-459` (`_get_audio_segment` trimming).  The arithmetic lives in
third-party `librosa==0.7.2` (`install/requirements.txt:6`), absent here; its
published algorithm for `librosa.stft(y, n_fft=511, hop_length=120, win_length=240,
window="hann", center=True, pad_mode="constant")` is restated:
  * window = periodic Hann(240) (scipy `get_window("hann", 240, fftbins=True)`),
    zero-padded centrally to 511 (135 left / 136 right);
  * signal zero-padded by n_fft//2 = 255 on both sides;
  * frame t = padded[t*120 : t*120+511], T = 1 + (len(y)-1)//120 frames;
  * X = rfft(window * frame) in float64 (256 bins), stored as complex64;
then the reference takes log(real(X*conj(X)) + 1e-6) in float32.
PARITY UNPINNED against librosa itself (not installable offline); pinned by the
known-answer tests in tests/test_stft_oracle.py (pure tone, Parseval, shapes).
"""
import numpy as np

N_FFT = 511


def hann_periodic(n):
    return 0.5 - 0.5 * np.cos(2.0 * np.pi * np.arange(n) / n)


def stft_window(win_length=240, n_fft=N_FFT):
    w = np.zeros(n_fft, dtype=np.float64)
    lpad = (n_fft - win_length) // 2
    w[lpad:lpad + win_length] = hann_periodic(win_length)
    return w


def stft_complex(sample, hop=120, win_length=240, n_fft=N_FFT):
    y = np.asarray(sample)
    ypad = np.pad(y, n_fft // 2, mode="constant")
    n_frames = 1 + (len(ypad) - n_fft) // hop
    idx = np.arange(n_fft)[:, None] + hop * np.arange(n_frames)[None, :]
    frames = ypad[idx]                                   # (n_fft, T), dtype of y
    win = stft_window(win_length, n_fft)[:, None]       # float64
    return np.fft.rfft(win * frames, axis=0).astype(np.complex64)


def log_power_spectrogram(sample, sampling_rate=24000, window_ms=10, step_ms=5, eps=1e-6):
    """(256, T) float32, dataset.py:483-495."""
    nperseg = int(round(window_ms * sampling_rate / 1e3))
    hop = int(round(step_ms * sampling_rate / 1e3))
    S = stft_complex(sample, hop=hop, win_length=nperseg)
    return np.log(np.real(S * np.conj(S)) + eps)


def trim_audio(aud_sample, frame_idx, audio_length, sampling_rate=24000, vid_fps=60):
    """dataset.py:439-451: the `audio_length`-second window centred on frame_idx/fps, clamped."""
    min_len = int(audio_length * sampling_rate)
    max_len = aud_sample.shape[0]
    if max_len < min_len:
        aud_sample = np.pad(aud_sample, (0, min_len - max_len))
    start_sec = float(frame_idx / vid_fps) - (audio_length / 2)
    start = int(max(0, start_sec * sampling_rate))
    if start + min_len > max_len:
        start = max_len - min_len
    return aud_sample[start:start + min_len], start_sec
