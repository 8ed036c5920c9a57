"""Throughput of the on-device visual input pipeline (tbn_frames_to_tensor) for one config-4 batch (32 clips x 3
segments: 96 RGB frames + 96 x 10 flow frames, 256x456 -> 224x224) next to the oracle (NumPy, what the reference does
per sample on the host).  HBM-bound: algorithmic bytes = uint8 source box read once + fp32 output written once."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from attention_based_tbn_amd.config import load_config
from attention_based_tbn_amd.core.dataset import get_transforms
from oracle import transform as otf

cfg = load_config([])
rng = np.random.RandomState(0)
B = 96
rgb = torch.from_numpy(rng.randint(0, 256, (B, 256, 456, 3)).astype(np.uint8)).cuda()
flow = torch.from_numpy(rng.randint(0, 256, (B * 10, 256, 456, 1)).astype(np.uint8)).cuda()
for mode in ("train", "test"):
    tf = get_transforms(cfg, ["RGB", "Flow"], mode)
    for name, x, c in (("RGB", rgb, 3), ("Flow", flow, 10)):
        np.random.seed(0)
        for _ in range(3): tf[name](x)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        reps = 20
        for _ in range(reps): out = tf[name](x)
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / reps
        alg = out.numel() * 4 + out.numel()      # fp32 written + (>=) one uint8 read per output element
        print(f"{mode:5s} {name:4s}: {ms*1e3:8.1f} us per {B} samples  -> {B/ms*1e3:9.0f} samples/s, {alg/ms/1e9:6.2f} TB/s algorithmic "
              f"(launch + RNG draws included)")
# host reference arithmetic (oracle) on a bounded sample
frames = [f for f in rgb[:8].cpu().numpy()]
t0 = time.perf_counter(); n = 0
while time.perf_counter() - t0 < 5:
    np.random.seed(n)
    otf.stack_totensor_normalize(otf.train_geometry(frames, 224, [1, 0.875, 0.75, 0.66]), "RGB", [0.408, 0.459, 0.502], [1, 1, 1])
    n += 1
dt = time.perf_counter() - t0
print(f"host NumPy oracle, RGB train pipeline: {8*n/dt:8.1f} samples/s on 1 core")
