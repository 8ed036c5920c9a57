"""Times the log-power STFT kernel at config-4 volume (96 waveforms x 30 695 samples -> 96 x 256 x 256) and at the
2.1 s window (50 400 samples -> 256 x 420): us per call, TFLOP/s against the fp32 MFMA peak (algorithmic FLOPs =
2 * 2 * 256 * 240 per frame), bytes moved (waveform in, spectrogram out) against HBM peak.  Under
`rocprofv3 --kernel-trace --stats -- python3 scripts/stft_profile.py` the same launches appear per kernel."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from attention_based_tbn_amd.core.dataset import Spectrogram
spec = Spectrogram()
for nseg, L in ((96, 30695), (192, 30695), (96, 50400)):
    wave = 0.1 * torch.randn(nseg, L, device="cuda")
    for _ in range(3):
        out = spec(wave)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    reps = 20
    e0.record()
    for _ in range(reps):
        out = spec(wave)
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / reps * 1e3
    W = out.shape[2]
    flops = 2.0 * 2 * 256 * 240 * W * nseg
    byt = wave.numel() * 4 + out.numel() * 4
    print(f"stft_logpower {nseg} x {L} samples -> {tuple(out.shape)}: {us:7.1f} us  {flops / us / 1e6:6.1f} TFLOP/s "
          f"({100 * flops / us / 1e6 / 157.3:4.1f} % of fp32 MFMA peak)  {byt / us / 1e6:5.2f} TB/s of in+out bytes")
