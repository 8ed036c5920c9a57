"""CPU: the train-step / metric oracle against the fixtures produced by torch's own optimiser and the
unmodified reference Metric / save_scores (tests/golden/make_golden_trainstep.py)."""
import json
import os

import numpy as np
import torch

from oracle import trainstep as ot

GOLD = os.path.join(os.path.dirname(__file__), "golden")
SIZES = 3


def test_sgd_clip_schedule_oracle_matches_torch():
    z = np.load(os.path.join(GOLD, "trainstep.npz"))
    p = [z["p0_%d" % i].astype(np.float32) for i in range(SIZES)]
    buf = [None] * SIZES
    for s in range(3):
        g = [z["g%d_%d" % (s, i)].astype(np.float32) for i in range(SIZES)]
        total, gc = ot.clip_grad_norm(g, 20)
        assert abs(total - z["norm%d" % s]) <= 2e-6 * z["norm%d" % s]
        if s == 1:
            for i in range(SIZES):
                np.testing.assert_allclose(gc[i], z["gclip1_%d" % i], rtol=2e-6, atol=0)
        lr = ot.multistep_lr(0.01, s, [2], 0.1)
        for i in range(SIZES):
            p[i], buf[i] = ot.sgd_step(p[i], gc[i], buf[i], lr, 0.9, 0.0005)
            np.testing.assert_allclose(p[i], z["p%d_%d" % (s + 1, i)], rtol=1e-6, atol=1e-7)
        assert abs(ot.multistep_lr(0.01, s + 1, [2], 0.1) - z["lr%d" % s]) < 1e-9
    for i in range(SIZES):
        np.testing.assert_allclose(buf[i], z["m3_%d" % i], rtol=2e-6, atol=1e-7)


def test_metric_oracle_matches_reference():
    doc = json.load(open(os.path.join(GOLD, "metric.json")))
    m = ot.Metric({"verb": 12, "noun": 17}, [1, 5], 2, extra_losses=("entropy",))
    for b in doc["batches"]:
        m.set_metrics({"verb": b["verb"], "noun": b["noun"]}, {"verb": b["t_verb"], "noun": b["t_noun"]}, b["B"],
                      b["loss"])
    loss, acc, cm = m.get_metrics()
    exp = doc["expected"]
    assert acc == exp["accuracy"]
    for k, v in exp["loss"].items():
        assert abs(loss[k] - v) <= 1.1e-5, (k, loss[k], v)
    for k in ("verb", "noun"):
        assert np.array_equal(cm[k], np.array(exp["conf_mat"][k], dtype=np.float32))


def test_save_scores_matches_reference_document(tmp_path):
    from attention_based_tbn_amd.core.utils.misc import save_scores
    doc = json.load(open(os.path.join(GOLD, "scores.json")))
    sc = {k: [torch.tensor(t) for t in v] for k, v in doc["input"].items()}
    fn = str(tmp_path / "out" / "scores.json")
    save_scores(sc, fn, doc["action_names"])
    got = json.load(open(fn))
    assert got == doc["expected"]
    with open(fn) as f:      # same formatting on the wire (indent=4, key order)
        assert f.read() == json.dumps(doc["expected"], indent=4)


def test_reference_parameter_names_and_optimizer_state_roundtrip():
    """host logic of the checkpoint format: flat GPU storage <-> the reference's per-layer optimizer slots"""
    from attention_based_tbn_amd.config import get_modality, load_config
    from attention_based_tbn_amd.core.models import build_model
    from attention_based_tbn_amd.core.utils.misc import (optimizer_state_from_reference, optimizer_state_to_reference,
                                                          reference_parameter_names)
    keys = json.load(open(os.path.join(GOLD, "keys_cfg3_rgb_audio_mha_T8.json")))
    cfg = load_config(keys["overrides"])
    model, _, _ = build_model(cfg, get_modality(cfg), torch.device("cpu"))
    names = reference_parameter_names(model)
    buffers = (".running_mean", ".running_var", ".num_batches_tracked")
    # the reference's parameters = its state_dict keys minus the buffers (BN statistics, the PE table)
    ref_params = [k for k, _ in keys["keys"] if not k.endswith(buffers) and not k.endswith(".pe")]
    assert names == ref_params
    assert set(keys["trainable"]) <= set(names)
    # a reference-style optimizer state: one momentum buffer per parameter, value = its index
    sd = model.state_dict()
    ref_state = {"state": {i: {"momentum_buffer": torch.full_like(sd[n], float(i))} for i, n in enumerate(names)},
                 "param_groups": [{"lr": 0.003, "momentum": 0.9, "weight_decay": 0.0005, "params": list(range(len(names)))}]}
    opt = torch.optim.SGD(model.parameters(), 0.01, momentum=0.9)
    optimizer_state_from_reference(model, opt, ref_state)
    assert opt.param_groups[0]["lr"] == 0.003
    back = optimizer_state_to_reference(model, opt)
    assert back["param_groups"][0]["params"] == list(range(len(names)))
    for i, n in enumerate(names):
        assert torch.equal(back["state"][i]["momentum_buffer"], ref_state["state"][i]["momentum_buffer"]), n


def test_attention_priors_match_reference():
    """gaussian / uniform / loud priors (dataset.py:534-575) on the generator's seeded spectrograms"""
    from attention_based_tbn_amd.core.dataset import attention_prior
    doc = json.load(open(os.path.join(GOLD, "priors.json")))
    rng = np.random.RandomState(4)
    kinds = set()
    for c in doc["cases"]:
        spec = rng.randn(16, c["W"]).astype(np.float32)
        spec[:, rng.randint(0, c["W"])] += 9.0
        got = attention_prior(spec, c["audio_length"], c["prior_type"])
        exp = torch.tensor(c["expected"], dtype=torch.float64).float()
        assert got.dtype == torch.float32 and got.shape == exp.shape
        assert torch.equal(got, exp), (c["prior_type"], c["audio_length"])
        kinds.add(c["prior_type"])
    assert kinds == {"gaussian", "uniform", "loud"}
