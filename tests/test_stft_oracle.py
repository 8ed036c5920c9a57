"""CPU: known-answer tests that pin the restated librosa.stft semantics (oracle/stft.py).
librosa itself is not installable offline -> "parity unpinned" against it; these
tests pin the documented algorithm (window, centring, frame count, bin mapping)."""
import numpy as np

from oracle.stft import hann_periodic, log_power_spectrogram, stft_complex, stft_window, trim_audio


def test_shapes_for_reference_lengths():
    # reference: W = 1 + (int(audio_length*24000) - 1)//120  (dataset.py:439,483-493)
    for sec, W in ((1.279, 256), (1.28, 256), (2.1, 420), (4.0, 800)):
        L = int(sec * 24000)
        S = log_power_spectrogram(np.zeros(L, dtype=np.float32))
        assert S.shape == (256, W) and S.dtype == np.float32
        assert np.allclose(S, np.log(np.float32(1e-6)))


def test_window_layout():
    w = stft_window()
    assert w.shape == (511,) and w[:135].sum() == 0 and w[375:].sum() == 0
    assert w[135] == 0.0 and np.isclose(w[135 + 120], 1.0)          # periodic hann: w[0]=0, peak at n/2
    assert np.allclose(w[135:375], hann_periodic(240))
    assert np.count_nonzero(w) == 239


def test_pure_tone_bin_and_amplitude():
    sr, L, k0 = 24000, 30695, 40
    f = k0 * sr / 511.0
    t = np.arange(L) / sr
    y = (0.5 * np.cos(2 * np.pi * f * t)).astype(np.float32)
    S = np.abs(stft_complex(y))
    mid = S[:, 20:-20]                                              # away from the zero-padded edges
    assert (mid.argmax(axis=0) == k0).all()
    # |X[k0]| = A/2 * sum(window) for an on-bin tone
    assert np.allclose(mid[k0], 0.25 * hann_periodic(240).sum(), rtol=2e-3)


def test_parseval_per_frame():
    rng = np.random.RandomState(0)
    y = rng.randn(30695).astype(np.float32) * 0.1
    X = stft_complex(y).astype(np.complex128)
    ypad = np.pad(y.astype(np.float64), 255)
    w = stft_window()
    for t in (0, 5, 100, 255):
        fr = w * ypad[t * 120: t * 120 + 511]
        # full-spectrum energy from the one-sided rfft of an odd-length frame
        e = np.abs(X[0, t]) ** 2 + 2 * (np.abs(X[1:, t]) ** 2).sum()
        assert np.isclose(e / 511.0, (fr ** 2).sum(), rtol=1e-5)


def test_trim_clamps_and_centres():
    a = np.arange(100000, dtype=np.float32)
    seg, start_sec = trim_audio(a, frame_idx=120, audio_length=1.279)   # 2.0 s centre
    assert len(seg) == 30695 and seg[0] == int((2.0 - 0.6395) * 24000)
    seg, _ = trim_audio(a, frame_idx=0, audio_length=1.279)
    assert seg[0] == 0
    seg, _ = trim_audio(a, frame_idx=60 * 100, audio_length=1.279)
    assert seg[-1] == 99999


# ---- spec_type "logms" (dataset.py:496-506); librosa absent: known-answer tests of the restated algorithm ------------
def test_mel_filterbank_known_answers():
    from oracle.stft import hz_to_mel, mel_filterbank, mel_to_hz
    from attention_based_tbn_amd.core.dataset.spectrogram import mel_filterbank as product_fb
    fb = mel_filterbank(24000, 511, 128)
    assert fb.shape == (128, 256) and fb.dtype == np.float32 and (fb >= 0).all()
    assert np.array_equal(fb, product_fb(24000, 511, 128))          # the product builds the same basis
    # Slaney scale: linear (200/3 Hz per mel) below 1 kHz, logarithmic above; exact inverse pair
    assert abs(hz_to_mel(1000.0) - 15.0) < 1e-12 and abs(hz_to_mel(500.0) - 7.5) < 1e-12
    assert abs(mel_to_hz(hz_to_mel(6400.0)) - 6400.0) < 1e-9 and abs(hz_to_mel(6400.0) - 42.0) < 1e-9
    # triangles: peaks move monotonically up in frequency, every filter is non-empty, neighbours overlap
    peaks = fb.argmax(1)
    assert (np.diff(peaks) >= 0).all() and peaks[0] <= 2 and peaks[-1] >= 245
    assert (fb.sum(1) > 0).all()
    # Slaney normalisation: each triangle has (nearly) unit area in Hz -> sum * bin width ~ 1 once it spans a few bins
    binw = 12000.0 / 255
    area = fb.sum(1) * binw
    assert np.all(np.abs(area[80:] - 1.0) < 0.05)      # narrow low-frequency triangles straddle 1-2 FFT bins


def test_log_mel_spectrogram_known_answers():
    from oracle.stft import log_mel_spectrogram, mel_filterbank
    sr, f0 = 24000, 3000.0
    t = np.arange(int(1.279 * sr)) / sr
    db = log_mel_spectrogram(np.sin(2 * np.pi * f0 * t).astype(np.float32))
    assert db.shape == (128, 256) and db.dtype == np.float32
    assert abs(db.max()) < 1e-6 and db.min() >= -80.0 - 1e-4            # relative to the maximum, clipped at -80 dB
    fb = mel_filterbank()
    want_bin = int(np.argmax(fb[:, int(round(f0 / (12000.0 / 255)))]))
    assert abs(int(np.argmax(db[:, 128])) - want_bin) <= 1              # the tone lands in its mel band
    # doubling the amplitude leaves a max-referenced dB map unchanged
    db2 = log_mel_spectrogram((2 * np.sin(2 * np.pi * f0 * t)).astype(np.float32))
    assert np.abs(db2 - db)[db > -70].max() < 1e-3


def test_restatement_agrees_with_scipy_stft():
    """independent implementation of the same framing: scipy.signal.stft with the 511-sample centrally padded periodic
    Hann window (scipy.signal.get_window("hann", 240, fftbins=True) is what librosa 0.7.2 itself calls), hop 120,
    zero boundary extension of n_fft // 2 -- librosa.stft's centre=True / pad_mode="constant" -- and no end padding"""
    import scipy.signal as ss
    rng = np.random.RandomState(3)
    for L in (30695, 30720, 50400, 1000):
        y = (0.1 * rng.randn(L)).astype(np.float32)
        w = np.zeros(511)
        w[135:375] = ss.get_window("hann", 240, fftbins=True)
        assert np.allclose(w, stft_window(), rtol=0, atol=1e-15)
        _, _, Z = ss.stft(y.astype(np.float64), window=w, nperseg=511, noverlap=511 - 120, nfft=511, boundary="zeros",
                          padded=False, return_onesided=True)
        Z = Z * w.sum()                                   # scipy normalises by the window sum
        S = stft_complex(y)
        assert S.shape == Z.shape == (256, 1 + (L - 1) // 120)
        assert np.abs(S - Z).max() < 1e-5 * np.abs(Z).max()
        ref = np.log((Z * np.conj(Z)).real + 1e-6).astype(np.float32)
        assert np.abs(log_power_spectrogram(y) - ref).max() < 2e-4      # complex64 storage in the restatement
