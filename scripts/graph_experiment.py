"""Experiment: capture the whole train step (fwd + loss + bwd + clip + SGD) of BASELINE config CFG (default 4; CFG=2: the lone
RGB backbone) in one HIP graph and replay it.  MULTI=0/1 modality streams, AUX=0/1 weight-gradient stream (not capturable
together with MULTI: nested forks, profiles/HISTORY.md), BRANCH=0/1 branch-level side stream of the eager run (a captured
step always runs the one-chain program with riders: core/models/bn_inception.py passes no side / aux stream while capturing)."""
import faulthandler, os, sys, time, torch
faulthandler.enable(all_threads=True)      # a SIGSEGV prints the Python stack of every thread (autograd thread included)
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from attention_based_tbn_amd.config import load_config, get_modality
from attention_based_tbn_amd.core.models import build_model
from attention_based_tbn_amd.core.utils import FusedSGD

dev = torch.device("cuda", 0)
C_ = bench.CONFIGS[int(os.environ.get("CFG", "4"))]
cfg = load_config(C_["ov"]); modality = get_modality(cfg)
torch.manual_seed(0)
model, criterion, _ = build_model(cfg, modality, dev); model.train()
params = [p for p in model.parameters() if p.requires_grad]
opt = FusedSGD(params, lr=cfg.train.optim.lr, momentum=cfg.train.optim.momentum, weight_decay=cfg.train.optim.weight_decay)
B, n = C_["batch"], 3
inp, tgt = bench.synthetic_batch(B, n, dev, 0, modality)
core = getattr(model, "module", model)
core.multi_stream = os.environ.get("MULTI", "1") == "1"
for m in modality:
    getattr(core, "Base_" + m).use_aux_stream = os.environ.get("AUX", "1") == "1"
    if "BRANCH" in os.environ:
        getattr(core, "Base_" + m).use_branch_streams = os.environ["BRANCH"] == "1"
print("multi_stream", core.multi_stream, "aux", os.environ.get("AUX", "1"), flush=True)

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
print("eager       %.3f ms/step" % timeit(step, 30), flush=True)
s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    for _ in range(3): step()
torch.cuda.current_stream().wait_stream(s)
g = torch.cuda.CUDAGraph()
opt.zero_grad(set_to_none=True)
# during the capture every backbone entry point announces itself on stderr before and after the call: if the process
# dies inside the capture the last line names the call (forward / backward of which plan), if it dies after the last
# "<-" the fault is in the end-of-capture processing
from attention_based_tbn_amd import _lib as _tl
from attention_based_tbn_amd.core.models import bn_inception as _bi
_orig_call = _tl.call
def _traced(name, *a):
    if name.startswith("tbn_backbone_"):
        print("[capture] -> %s plan=%x" % (name, getattr(a[0], "value", 0) or 0), file=sys.stderr, flush=True)
    _orig_call(name, *a)
    if name.startswith("tbn_backbone_"):
        print("[capture] <- %s" % name, file=sys.stderr, flush=True)
_bi.call = _traced
try:
    with torch.cuda.graph(g):
        static_loss = step()
        print("[capture] step() returned: leaving the capture context (hipStreamEndCapture)", file=sys.stderr, flush=True)
    _bi.call = _orig_call
    print("captured", flush=True)
    for _ in range(3): g.replay()
    print("graph replay %.3f ms/step  loss %.4f" % (timeit(g.replay, 30), static_loss.item()), flush=True)
except Exception as e:
    print("capture failed:", type(e).__name__, str(e)[:400])
