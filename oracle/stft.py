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
known-answer tests in tests/test_stft_oracle.py (pure tone, Parseval, shapes) and
cross-checked there against scipy.signal.stft, an independent implementation of
the same framing (zero boundary extension by n_fft // 2, hop 120, no end padding).
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


# ---------------------------------------------------------------------------------------------------
# spec_type "logms" (reference dataset.py:496-506): librosa.feature.melspectrogram(sample, sr, n_fft=511,
# window="hann", hop_length, win_length, pad_mode="constant") followed by librosa.power_to_db(S, ref=np.max).
# librosa 0.7.2's published algorithm, restated (PARITY UNPINNED, librosa is not installable offline):
#   S = |stft|^2 (power = 2.0);  mel basis = librosa.filters.mel(sr, n_fft, n_mels=128, fmin=0, fmax=sr/2,
#   htk=False, norm=1 "slaney"): Slaney mel scale (linear below 1 kHz, log above), triangular filters on the FFT
#   bin centres, each scaled by 2 / (f[i+2] - f[i]);  power_to_db: 10 log10(max(S, 1e-10)) - 10 log10(max(1e-10,
#   ref)), then clipped at (max - 80 dB).
def hz_to_mel(f):
    f = np.asarray(f, dtype=np.float64)
    f_sp = 200.0 / 3
    mels = f / f_sp
    min_log_hz, min_log_mel, logstep = 1000.0, 1000.0 / f_sp, np.log(6.4) / 27.0
    return np.where(f >= min_log_hz, min_log_mel + np.log(np.maximum(f, 1e-30) / min_log_hz) / logstep, mels)


def mel_to_hz(m):
    m = np.asarray(m, dtype=np.float64)
    f_sp = 200.0 / 3
    min_log_hz, min_log_mel, logstep = 1000.0, 1000.0 / f_sp, np.log(6.4) / 27.0
    return np.where(m >= min_log_mel, min_log_hz * np.exp(logstep * (m - min_log_mel)), f_sp * m)


def mel_filterbank(sr=24000, n_fft=N_FFT, n_mels=128):
    """(n_mels, 1 + n_fft//2) float32, Slaney-normalised"""
    fmax = sr / 2.0
    fftfreqs = np.linspace(0, fmax, 1 + n_fft // 2, endpoint=True)
    mel_f = mel_to_hz(np.linspace(hz_to_mel(0.0), hz_to_mel(fmax), n_mels + 2))
    fdiff = np.diff(mel_f)
    ramps = mel_f[:, None] - fftfreqs[None, :]
    w = np.zeros((n_mels, 1 + n_fft // 2))
    for i in range(n_mels):
        lower = -ramps[i] / fdiff[i]
        upper = ramps[i + 2] / fdiff[i + 1]
        w[i] = np.maximum(0, np.minimum(lower, upper))
    w *= (2.0 / (mel_f[2:n_mels + 2] - mel_f[:n_mels]))[:, None]
    return w.astype(np.float32)


def log_mel_spectrogram(sample, sampling_rate=24000, window_ms=10, step_ms=5, n_mels=128, top_db=80.0):
    """(128, T) float32 in dB relative to the maximum"""
    nperseg = int(round(window_ms * sampling_rate / 1e3))
    hop = int(round(step_ms * sampling_rate / 1e3))
    S = stft_complex(sample, hop=hop, win_length=nperseg)
    power = (np.abs(S) ** 2).astype(np.float32)
    mel = mel_filterbank(sampling_rate, N_FFT, n_mels).dot(power)
    amin = 1e-10
    ref = np.max(mel)
    db = 10.0 * np.log10(np.maximum(amin, mel)) - 10.0 * np.log10(np.maximum(amin, ref))
    return np.maximum(db, db.max() - top_db).astype(np.float32)
