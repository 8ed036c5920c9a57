"""GPU: branch mode of the backbone engine (tbn_backbone_params.side_stream): the 3x3 / pool_proj chain of every
inception block on a side stream beside the 1x1 -> double_3x3 chain.  The branches only meet at the concat (reference
core/models/bn_inception_audio.py:437-1003, torch.cat :485-493), so the two-stream program must reproduce the serial
one BIT FOR BIT once both launch the same kernels -- after tuning, the sibling-pair decisions are taken out of the plan
(`_Plan.clear_pairs`: an edit of the exported plan blob, tbn_backbone_plan_export / _import) so that the serial program
does not merge 3x3 | double_3x3_1 into one launch (other tiles, other partial-sum order): what is left to differ is only
what a race between the two chains would break (shared scratch slots, a missing join).  (Rounds 4-5 used the A/B knob
TBN_USE_PAIRS=0 for this; the shipped library reads no environment variable any more.)"""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

_WORKER = r'''
import sys, torch
sys.path.insert(0, sys.argv[1])
from attention_based_tbn_amd._lib import call, lib
from attention_based_tbn_amd.core.models.bn_inception import BNInception
DEV = torch.device("cuda")
for cin, N, H, W in ((3, 6, 224, 224), (1, 5, 128, 256), (10, 3, 96, 96)):
    torch.manual_seed(cin)
    net = BNInception(1000, cin).to(DEV)
    with torch.no_grad():
        net.running_var.uniform_(0.5, 1.5); net.running_mean.normal_(0, 0.1)
    net.set_bn_trainable(True, True)
    x = torch.randn(N, cin, H, W, device=DEV)
    rm0, rv0 = net.running_mean.clone(), net.running_var.clone()

    def train_step(branch, aux):
        net.train(); net.use_branch_streams = branch; net.use_aux_stream = aux
        net.running_mean.copy_(rm0); net.running_var.copy_(rv0)
        net.zero_grad(set_to_none=True)
        out = net(x)
        (out.square().mean() + out.sum() * 1e-3).backward()
        torch.cuda.synchronize()
        return [out.detach().clone(), net.flat_weight.grad.clone(), net.flat_bias.grad.clone(), net.bn_weight_first.grad.clone(),
                net.bn_weight_rest.grad.clone(), net.bn_bias_rest.grad.clone(), net.running_mean.clone(), net.running_var.clone()]

    def eval_fwd(branch):
        net.eval(); net.use_branch_streams = branch
        net.running_mean.copy_(rm0); net.running_var.copy_(rv0)
        with torch.no_grad():
            return net(x).clone()

    train_step(False, False)                           # serial program, one stream: autotunes on first use ...
    plan = net._plans[(N, H, W)]
    plan.clear_pairs()                                 # ... then every pair runs as its two tuned single launches
    ref = train_step(False, False)
    assert lib().tbn_backbone_num_streams(plan.handle) == 2
    assert float(ref[1].abs().max()) > 0 and all(torch.isfinite(t).all() for t in ref)
    for rep in range(3):                               # repeated: a race need not show on the first try
        for branch, aux in ((True, True), (True, False), (False, True)):
            got = train_step(branch, aux)
            for i, (a, b) in enumerate(zip(got, ref)):
                assert torch.equal(a, b), (cin, rep, branch, aux, i, float((a - b).abs().max()))
    eval_fwd(False)                                    # eval-mode tuning
    plan.clear_pairs()
    e0 = eval_fwd(False)
    for rep in range(3):
        assert torch.equal(eval_fwd(True), e0), (cin, "eval", rep)
    # a side stream inside a capture is ignored (serial program): covered by tests/test_model_gpu.py's capture tests
print("BRANCH_OK")
'''


def test_branch_mode_is_bit_identical_to_the_serial_program(tmp_path):
    script = tmp_path / "branch_worker.py"
    script.write_text(_WORKER)
    env = dict(os.environ)
    r = subprocess.run([sys.executable, str(script), ROOT], env=env, cwd=ROOT, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and "BRANCH_OK" in r.stdout, (r.stdout[-1500:] + "\n----\n" + r.stderr[-3000:])
