"""Shared helpers for the parity tests (oracle side = checker only)."""
import json
import os

import numpy as np
import torch

from attention_based_tbn_amd.config import load_config, get_modality

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load_case(name):
    with open(os.path.join(GOLDEN, f"keys_{name}.json")) as f:
        meta = json.load(f)
    data = dict(np.load(os.path.join(GOLDEN, f"model_{name}.npz")))
    cfg = load_config(meta["overrides"])
    modality = get_modality(cfg)
    assert modality == meta["modality"]
    inp = {k[3:]: torch.from_numpy(v.astype(np.float32)) for k, v in data.items() if k.startswith("in_")}
    target = {"class": {k[4:]: torch.from_numpy(v) for k, v in data.items() if k.startswith("tgt_")}}
    return cfg, modality, meta, data, inp, target


def prior_target(cfg, B, n):
    from oracle.tbn import gaussian_kernel
    T = round(cfg.data.audio.audio_length * 25 / 4)
    return torch.from_numpy(gaussian_kernel(T, 1)).float().view(1, 1, T, 1).repeat(B, n, 1, 1)


def build_oracle(cfg, modality, meta):
    """Oracle model with the golden case's name-keyed weights; checks key/shape contract."""
    from oracle.fill import fill_state_dict, pretrained_pair
    from oracle.tbn import build_model
    model, crit, _ = build_model(cfg, modality, pretrained_pair(7))
    sd = model.state_dict()
    assert [[k, list(v.shape)] for k, v in sd.items()] == meta["keys"]
    model.load_state_dict(fill_state_dict(sd, meta["fill_seed"]))
    return model, crit


def rel_err(a, b):
    a = torch.as_tensor(a).detach().double()
    b = torch.as_tensor(b).detach().double()
    return float((a - b).abs().max() / (b.abs().max() + 1e-30))


def assert_close(got, want, what="", rtol=1e-3, atol_frac=1e-5):
    """the north star's "within 1e-3 relative fp32 tolerance", asserted BOTH ways (round-5 verdict): the norm bound
    max|got - want| / max|want| < rtol (`rel_err`) and element-wise |got - want| <= rtol * |want| + atol_frac * max|want|
    (torch.allclose with an absolute floor scaled to the tensor: an element near zero may be off by 1e-5 of the largest)."""
    a = torch.as_tensor(got).detach().double().cpu()
    b = torch.as_tensor(want).detach().double().cpu()
    assert a.shape == b.shape, (what, tuple(a.shape), tuple(b.shape))
    e = rel_err(a, b)
    assert e < rtol, (what, "norm bound", e)
    atol = atol_frac * float(b.abs().max())
    bad = (a - b).abs() > rtol * b.abs() + atol
    assert not bool(bad.any()), (what, "element-wise", int(bad.sum()), float(((a - b).abs() - rtol * b.abs()).max()), atol)
    return e
