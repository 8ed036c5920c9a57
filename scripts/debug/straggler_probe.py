"""debug: what happens in the one slow step of a config-3 run (93 - 137 ms against 50)?  Per step: GPU time, device
allocations of the caching allocator, Python GC collections."""
import gc, os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench
from attention_based_tbn_amd.config import load_config, get_modality
from attention_based_tbn_amd.core.models import build_model
from attention_based_tbn_amd.core.utils import FusedSGD
cfgno = int(sys.argv[1]) if len(sys.argv) > 1 else 3
dev = torch.device("cuda", 0)
C_ = bench.CONFIGS[cfgno]
cfg = load_config(C_["ov"]); modality = get_modality(cfg)
torch.manual_seed(0)
model, criterion, _ = build_model(cfg, modality, dev); model.train()
opt = FusedSGD([p for p in model.parameters() if p.requires_grad], lr=0.01, momentum=0.9)
B = C_["batch"]
inp, tgt = bench.synthetic_batch(B, 3, dev, 0, modality)
if model.use_attention:
    pass
gcs = []
gc.callbacks.append(lambda phase, info: gcs.append((time.perf_counter(), phase, info.get("generation"), info.get("collected"))))
def step():
    opt.zero_grad(set_to_none=True)
    out = model(inp)
    loss, _ = model.get_loss(criterion, tgt, out, 0)
    loss["total"].backward()
    opt.step(clip_grad=20, grads_consumed=True)
for _ in range(6): step()
torch.cuda.synchronize()
N = 60
marks = [torch.cuda.Event(enable_timing=True) for _ in range(N + 1)]
stats, host = [], []
for i in range(N):
    marks[i].record()
    t0 = time.perf_counter()
    step()
    host.append(time.perf_counter() - t0)
    s = torch.cuda.memory_stats()
    stats.append((s["num_device_alloc"], s["num_device_free"], s["num_alloc_retries"], s["reserved_bytes.all.current"] >> 20))
marks[N].record()
torch.cuda.synchronize()
ms = [marks[i].elapsed_time(marks[i + 1]) for i in range(N)]
med = sorted(ms)[N // 2]
print("median %.2f ms" % med)
for i in range(N):
    flag = " <== slow" if ms[i] > 1.15 * med else ""
    d = tuple(a - b for a, b in zip(stats[i][:3], stats[i - 1][:3])) if i else (0, 0, 0)
    if flag or d != (0, 0, 0) or host[i] > 0.03:
        print("step %2d gpu %.2f ms host %.1f ms  device alloc/free/retries in this step %s reserved %d MiB%s" % (i, ms[i], host[i] * 1e3, d, stats[i][3], flag))
print("gc events (generation 2 only):", [(round(t, 3), ph, g, c) for t, ph, g, c in gcs if g == 2][:10])
