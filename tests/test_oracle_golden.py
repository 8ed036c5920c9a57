"""CPU: the oracle restatement against fixtures produced by the unmodified reference
(tests/golden/make_golden.py).  This is what pins the oracle."""
import json
import os

import numpy as np
import pytest
import torch

from tests.util import GOLDEN, build_oracle, load_case, prior_target, rel_err

EVAL_CASES = ["cfg1_audio_only", "cfg2_rgb_only", "cfg3_rgb_audio_mha_T8", "cfg3_rgb_audio_mha_T13",
              "cfg4_all_noattn", "cfg5_all_mha_eval", "fixed_attn", "unimodal_attn", "proto_attn"]
TRAIN_CASES = ["train_cfg4_all_noattn", "train_cfg3_mha"]
TOL = 1e-5  # same torch ops on the same CPU: differences are summation-order only


def test_trunk_matches_in_repo_graph():
    """oracle trunk == reference core/models/bn_inception_audio.py:446-1003 on its own pool1 output"""
    from oracle.bninception import BNInception
    from oracle.fill import fill_state_dict
    d = np.load(os.path.join(GOLDEN, "trunk_pin.npz"))
    net = BNInception(1000, 1)
    net.load_state_dict(fill_state_dict(net.state_dict(), int(d["seed"])))
    for mode in ("eval", "train"):
        net.train(mode == "train")
        with torch.no_grad():
            feat = net.trunk(torch.from_numpy(d[mode + "_p1"]))
            net.is_audio, net.attend = True, True
            freq = net.logits(feat)
            net.attend = False
            glob = net.logits(feat)
        assert rel_err(feat, d[mode + "_feat"]) < TOL
        assert rel_err(freq, d[mode + "_logits_freq"]) < TOL
        assert rel_err(glob, d[mode + "_logits_global"]) < TOL


def test_factory_audio_first_conv():
    """reference bn_inception.py:38-107: audio conv1 = channel-mean of the RGB conv1; last_linear dropped"""
    from oracle.bninception import bninception
    from oracle.fill import pretrained_pair
    d = np.load(os.path.join(GOLDEN, "factory_audio.npz"))
    pre = pretrained_pair(int(d["seed"]))
    m = bninception(1, "Audio", pre["imagenet"], is_audio=True, attend=True)
    assert not hasattr(m, "last_linear") and not bool(d["has_last_linear"])
    assert torch.equal(m.conv1_7x7_s2.weight.detach(), torch.from_numpy(d["conv1_w"]))
    m.eval()
    with torch.no_grad():
        y = m(torch.from_numpy(d["x"].astype(np.float32)))
    assert y.shape == (1, 1024, 1, 8)
    assert rel_err(y, d["y"]) < TOL
    mf = bninception(10, "Flow", pre["kinetics"], num_classes=400)
    assert abs(float(mf.conv1_7x7_s2.weight.detach().double().sum()) - float(d["flow_conv1_w_sum"])) < 1e-9


@pytest.mark.parametrize("name", EVAL_CASES)
def test_eval_forward_and_loss(name):
    cfg, modality, meta, data, inp, target = load_case(name)
    model, crit = build_oracle(cfg, modality, meta)
    model.eval()
    with torch.no_grad():
        out = model({k: v.clone() for k, v in inp.items()})
    for k, v in out.items():
        assert v.shape == data["out_" + k].shape
        assert rel_err(v, data["out_" + k]) < TOL, k
    for ep in (0, 20):
        loss, bs = model.get_loss(crit, target, out, epoch=ep)
        for k, v in loss.items():
            assert abs(float(v) - float(data[f"loss_ep{ep}_{k}"])) < 1e-5 * max(1, abs(float(v))), (ep, k)


@pytest.mark.parametrize("name", TRAIN_CASES)
def test_train_forward_loss_grads(name):
    cfg, modality, meta, data, inp, target = load_case(name)
    model, crit = build_oracle(cfg, modality, meta)
    assert [k for k, p in model.named_parameters() if p.requires_grad] == meta["trainable"]
    sd = {k: v.clone() for k, v in model.state_dict().items()}
    B, n = inp[modality[0]].shape[:2]
    if cfg.model.attention.enable and cfg.model.attention.use_prior:
        target["weights"] = prior_target(cfg, B, n)
    model.train()
    for ep in (0, 20):
        model.load_state_dict(sd)
        model.zero_grad()
        out = model({k: v.clone() for k, v in inp.items()})
        loss, bs = model.get_loss(crit, target, out, epoch=ep)
        loss["total"].backward()
        for k, v in out.items():
            assert rel_err(v, data[f"ep{ep}_out_{k}"]) < TOL, k
        for k, v in loss.items():
            v = float(v.detach()) if torch.is_tensor(v) else float(v)
            assert abs(v - float(data[f"ep{ep}_loss_{k}"])) < 1e-5 * max(1, abs(v)), (ep, k)
        params = dict(model.named_parameters())
        n_checked = 0
        for k in data:
            if k.startswith(f"ep{ep}_grad_"):
                assert rel_err(params[k[len(f"ep{ep}_grad_"):]].grad, data[k]) < 2e-4, k
                n_checked += 1
        assert n_checked >= (3 if ep == 0 else 1)
        gn = torch.sqrt(sum((p.grad.double() ** 2).sum() for p in model.parameters() if p.grad is not None))
        assert abs(float(gn) - float(data[f"ep{ep}_gradnorm"])) < 2e-4 * float(gn)
    st = model.state_dict()
    for k in data:
        if k.startswith("post_"):
            assert rel_err(st[k[5:]].double(), data[k]) < TOL, k


def test_audio_dropout_branches():
    """row a11, reference model.py:215-222: training, M > 1, data.audio.dropout = 0.5 -- the audio feature is zeroed when
    the host draw np.random.uniform() EXCEEDS the dropout value (inverted test), the audio backbone runs either way"""
    cfg, modality, meta, data, inp, target = load_case("train_audio_dropout")
    model, crit = build_oracle(cfg, modality, meta)
    sd = {k: v.clone() for k, v in model.state_dict().items()}
    model.train()
    assert set(meta["np_seeds"]) == {"drop", "keep"}
    for branch, seed in meta["np_seeds"].items():
        model.load_state_dict(sd)
        model.zero_grad()
        np.random.seed(seed)
        out = model({k: v.clone() for k, v in inp.items()})
        loss, _ = model.get_loss(crit, target, out, epoch=0)
        loss["total"].backward()
        for k, v in out.items():
            assert rel_err(v, data[f"{branch}_out_{k}"]) < TOL, (branch, k)
        for k, v in loss.items():
            v = float(torch.as_tensor(v).detach())
            assert abs(v - float(data[f"{branch}_loss_{k}"])) < 1e-5 * max(1, abs(v)), (branch, k)
        params = dict(model.named_parameters())
        for k in data:
            if k.startswith(f"{branch}_hasgrad_"):
                name = k[len(f"{branch}_hasgrad_"):]
                assert (params[name].grad is not None) == bool(data[k]), (branch, name)
                if bool(data[k]):
                    assert rel_err(params[name].grad, data[f"{branch}_grad_{name}"]) < 2e-4, (branch, name)
        key = "Base_Audio.conv1_7x7_s2_bn.running_mean"
        assert rel_err(model.state_dict()[key], data[f"{branch}_post_{key}"]) < TOL
    assert not bool(data["drop_hasgrad_Base_Audio.conv1_7x7_s2.weight"])    # dropped branch: no audio gradient


def test_crop_repeat_eval():
    """row a11, reference model.py:243-248: RGB carries k x the audio rows -> audio feature tiled k times, n *= k"""
    cfg, modality, meta, data, inp, target = load_case("crop_repeat_eval")
    model, crit = build_oracle(cfg, modality, meta)
    assert inp["RGB"].shape[1] == meta["repeat"] * inp["Audio"].shape[1]
    model.eval()
    with torch.no_grad():
        out = model({k: v.clone() for k, v in inp.items()})
    for k, v in out.items():
        assert v.shape == data["out_" + k].shape and rel_err(v, data["out_" + k]) < TOL, k
    loss, bs = model.get_loss(crit, target, out, epoch=0)
    for k, v in loss.items():
        assert abs(float(v) - float(data[f"loss_ep0_{k}"])) < 1e-5 * max(1, abs(float(v))), k


def test_sampler_bit_exact():
    """oracle sampler vs reference Video_Dataset.__getitem__ index selection (dataset.py:155-239)"""
    from oracle.sampler import sample_indices
    with open(os.path.join(GOLDEN, "sampler.json")) as f:
        g = json.load(f)
    assert any(r["stop_frame"] - r["start_frame"] < 8 for r in g["rows"])  # edge rows present
    for case in g["cases"]:
        np.random.seed(case["seed"])
        for row, want in zip(g["rows"], case["indices"]):
            got = sample_indices(row["start_frame"], row["stop_frame"], case["modality"], case["sampling"],
                                 case["mode"], case["num_segments"])
            for m in case["modality"]:
                assert got[m].dtype == np.int64
                assert got[m].tolist() == want[m], (case["mode"], case["sampling"], m, row)
