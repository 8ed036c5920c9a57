"""Times the log-power STFT kernel at config-4 volume (96 waveforms x 30 695 samples -> 96 x 256 x 256) and at the
2.1 s window (50 400 samples -> 256 x 420): us per call, TFLOP/s against the fp32 MFMA peak (algorithmic FLOPs =
2 * 2 * 256 * 240 per frame), bytes moved (waveform in, spectrogram out) against HBM peak.  Under
`rocprofv3 --kernel-trace --stats -- python3 scripts/stft_profile.py` the same launches appear per kernel.

Two numbers per shape: "cold" = 20 launches right after an idle GPU (a 1-ms burst), "warm" = 200 launches behind 30 ms
of the same launches.  Under MFMA load the shader clock of this part starts near 2.1 GHz and takes ~10 ms of sustained
load to reach 2.4 GHz (scripts/ubench/mfma_ramp.hip), so a short burst measures ~12 % low; the warm figure is the one
comparable with the training step's kernels (which run inside a continuously loaded GPU) and with the 2.4-GHz peak."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from attention_based_tbn_amd.core.dataset import Spectrogram
spec = Spectrogram()
for nseg, L in ((96, 30695), (192, 30695), (96, 50400)):
    wave = 0.1 * torch.randn(nseg, L, device="cuda")
    def timed(warm, reps):
        for _ in range(warm):
            out = spec(wave)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            out = spec(wave)
        e1.record()
        torch.cuda.synchronize()
        return out, e0.elapsed_time(e1) / reps * 1e3
    torch.cuda.synchronize()
    time.sleep(0.05)
    out, cold = timed(3, 20)
    out, us = timed(int(30e3 / cold) + 1, 200)
    W = out.shape[2]
    flops = 2.0 * 2 * 256 * 240 * W * nseg
    byt = wave.numel() * 4 + out.numel() * 4
    print(f"stft_logpower {nseg} x {L} samples -> {tuple(out.shape)}: warm {us:7.1f} us  {flops / us / 1e6:6.1f} TFLOP/s "
          f"({100 * flops / us / 1e6 / 157.3:4.1f} % of fp32 MFMA peak)  {byt / us / 1e6:5.2f} TB/s of in+out bytes;  "
          f"cold burst {cold:7.1f} us ({100 * flops / cold / 1e6 / 157.3:4.1f} %)")
