"""Whole train step of a BASELINE config captured in ONE HIP graph and replayed: under `rocprofv3 --kernel-trace` the eager
multi-stream step is host-bound (every launch is intercepted: ~40 us each, 1059 launches = 42 ms, the modality streams no
longer overlap), a graph replay is not -- its kernel trace shows the concurrency the un-traced step really has.
Usage: rocprofv3 --kernel-trace ... -- python3 scripts/graph_trace.py [config=4] [replays=6];  then scripts/step_timeline.py."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from attention_based_tbn_amd.config import load_config, get_modality
from attention_based_tbn_amd.core.models import build_model
from attention_based_tbn_amd.core.utils import FusedSGD

cfgno = int(sys.argv[1]) if len(sys.argv) > 1 else 4
replays = int(sys.argv[2]) if len(sys.argv) > 2 else 6
dev = torch.device("cuda", 0)
C_ = bench.CONFIGS[cfgno]
cfg = load_config(C_["ov"]); modality = get_modality(cfg)
torch.manual_seed(0)
model, criterion, _ = build_model(cfg, modality, dev); model.train()
params = [p for p in model.parameters() if p.requires_grad]
opt = FusedSGD(params, lr=cfg.train.optim.lr, momentum=cfg.train.optim.momentum, weight_decay=cfg.train.optim.weight_decay)
B, n = C_["batch"], 3
inp, tgt = bench.synthetic_batch(B, n, dev, 0, modality)

def step():
    opt.zero_grad(set_to_none=True)
    out = model(inp)
    loss, _ = model.get_loss(criterion, tgt, out, 0)
    loss["total"].backward()
    opt.step(clip_grad=cfg.train.clip_grad, grads_consumed=True)
    return loss["total"]

def timeit(fn, reps):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / reps * 1e3

for _ in range(5): step()
print("eager        %.3f ms/step" % timeit(step, 20), flush=True)
s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    for _ in range(3): step()
torch.cuda.current_stream().wait_stream(s)
g = torch.cuda.CUDAGraph()
opt.zero_grad(set_to_none=True)
with torch.cuda.graph(g):
    static_loss = step()
for _ in range(3): g.replay()
print("graph replay %.3f ms/step  loss %.4f" % (timeit(g.replay, replays), static_loss.item()), flush=True)
