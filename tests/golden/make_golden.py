#!/usr/bin/env python
"""Generate the golden fixtures under tests/golden/ by running the UNMODIFIED reference.

Runs only in the build container (needs /root/reference).  The reference's Python is
imported as is; the packages it needs but the image lacks are replaced by local stubs
injected into sys.modules:
  cv2 (INTER_LINEAR, getGaussianKernel), torchvision(.models/.transforms), librosa,
  pretrainedmodels.models.bninception.BNInception  (= oracle.bninception.BNInception,
  the restated third-party graph -- see oracle/__init__.py on what that does and does
  not pin),
and `torch.load` inside core.models.bn_inception is patched to return seeded state
dicts in place of the Drive-hosted .pth files.

Outputs (all small .npz, inputs stored as fp16-exact fp32):
  trunk_pin.npz      reference core/models/bn_inception_audio.py trunk vs oracle trunk
  factory_audio.npz  reference bninception() factory, Audio first-conv surgery
  model_*.npz        reference build_model/TBNModel forward (+loss, +grads) per case
  keys_*.json        state_dict key/shape lists (checkpoint-compat contract)
  sampler.json       reference Video_Dataset index sampling on annotation rows
  model_train_audio_dropout.npz / model_crop_repeat_eval.npz   row a11: audio-dropout branches, crop repeat
Usage:  python tests/golden/make_golden.py [trunk] [factory] [models] [a11] [sampler]   (no argument: everything)
"""
import importlib.util
import json
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"
sys.path.insert(0, ROOT)

from oracle import bninception as obn  # noqa: E402
from oracle.fill import fill_state_dict, pretrained_pair  # noqa: E402
from oracle.tbn import gaussian_kernel  # noqa: E402
from attention_based_tbn_amd.config import load_config  # noqa: E402


def install_stubs():
    # idempotent: reference modules imported after the first call hold references to THESE stub modules; a second call
    # (make_golden_trainstep.py runs several generators in one process) must not replace them, or attributes set later
    # (cv2.resize) would land on modules the reference code never sees
    if getattr(sys.modules.get("cv2"), "_tbn_stub", False):
        return
    cv2 = types.ModuleType("cv2")
    cv2._tbn_stub = True
    cv2.INTER_LINEAR = 1
    cv2.getGaussianKernel = lambda n, sigma: gaussian_kernel(n, sigma)
    sys.modules["cv2"] = cv2
    tv = types.ModuleType("torchvision")
    tv.models = types.ModuleType("torchvision.models")
    tv.transforms = types.ModuleType("torchvision.transforms")
    tv.transforms.functional = types.ModuleType("torchvision.transforms.functional")
    for name, mod in [("torchvision", tv), ("torchvision.models", tv.models),
                      ("torchvision.transforms", tv.transforms),
                      ("torchvision.transforms.functional", tv.transforms.functional)]:
        sys.modules[name] = mod
    sys.modules["librosa"] = types.ModuleType("librosa")
    ptm = types.ModuleType("pretrainedmodels")
    ptm.models = types.ModuleType("pretrainedmodels.models")
    ptm.models.bninception = types.ModuleType("pretrainedmodels.models.bninception")
    ptm.models.bninception.BNInception = obn.BNInception
    for name, mod in [("pretrainedmodels", ptm), ("pretrainedmodels.models", ptm.models),
                      ("pretrainedmodels.models.bninception", ptm.models.bninception)]:
        sys.modules[name] = mod
    sys.path.insert(0, REF)


def h16(t):
    """round to fp16-representable fp32 so fixtures store inputs exactly in half the bytes"""
    return t.half().float()


def save(name, **arrs):
    out = {}
    for k, v in arrs.items():
        if isinstance(v, torch.Tensor):
            v = v.detach().cpu().numpy()
        out[k] = v
    np.savez_compressed(os.path.join(HERE, name), **out)
    print("wrote", name, {k: getattr(v, "shape", None) for k, v in out.items()})


# ---------------------------------------------------------------------------------------------
def trunk_pin():
    """Pins the oracle's graph from conv2_3x3_reduce on against the in-repo statement."""
    spec = importlib.util.spec_from_file_location("ref_bna", REF + "/core/models/bn_inception_audio.py")
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    torch.manual_seed(5)
    ref = mod.BNInception_Audio(num_classes=1000, attend=False)
    ora = obn.BNInception(1000, 1)
    sd = fill_state_dict(ora.state_dict(), 5)
    ora.load_state_dict(sd)
    shared = {k: v for k, v in sd.items() if k in ref.state_dict()}
    missing = [k for k in ref.state_dict() if k not in shared]
    assert all(k.startswith("conv1_") for k in missing), missing
    ref.load_state_dict(shared, strict=False)
    res = {}
    for mode in ("eval", "train"):
        ref.train(mode == "train")
        taps = {}
        h = ref.pool1_3x3_s2.register_forward_hook(lambda m, i, o: taps.__setitem__("p1", o.detach().clone()))
        x = h16(torch.randn(2, 1, 96, 96, generator=torch.Generator().manual_seed(6)))
        with torch.no_grad():
            feat = ref.features(x)
            ref.attend = True
            freq = ref.logits(feat)
            ref.attend = False
            glob = ref.logits(feat)
        h.remove()
        res[mode + "_p1"] = taps["p1"]
        res[mode + "_feat"] = feat
        res[mode + "_logits_freq"] = freq
        res[mode + "_logits_global"] = glob
    save("trunk_pin.npz", seed=np.int64(5), **res)


def factory_audio():
    import core.models.bn_inception as rbi
    pre = pretrained_pair(3)
    orig = rbi.torch.load
    rbi.torch.load = lambda f, map_location=None: {k: v.clone() for k, v in
                                                   pre["kinetics" if "kinetics" in f else "imagenet"].items()}
    try:
        m = rbi.bninception(1, "Audio", pretrained="imagenet", model_dir="", is_audio=True, attend=True)
        mf = rbi.bninception(10, "Flow", pretrained="kinetics", model_dir="", is_audio=False, attend=True)
    finally:
        rbi.torch.load = orig
    x = h16(torch.randn(1, 1, 64, 256, generator=torch.Generator().manual_seed(4)))
    m.eval()
    with torch.no_grad():
        y = m(x)
    save("factory_audio.npz", seed=np.int64(3), conv1_w=m.conv1_7x7_s2.weight, x=x.half(), y=y,
         flow_conv1_w_sum=mf.conv1_7x7_s2.weight.double().sum(), has_last_linear=np.bool_(hasattr(m, "last_linear")))


# ---------------------------------------------------------------------------------------------
CASES = {
    # name: (overrides, modalities, B, n, (h, w) visual, audio H, mode, extras)
    "cfg1_audio_only": (["data.rgb.enable=False", "data.flow.enable=False", "model.attention.enable=False",
                         "data.audio.audio_length=1.279"], 2, 1, 64, 64, "eval"),
    "cfg2_rgb_only": (["data.flow.enable=False", "data.audio.enable=False", "model.attention.enable=False"],
                      2, 3, 64, 64, "eval"),
    "cfg3_rgb_audio_mha_T8": (["data.flow.enable=False", "data.audio.audio_length=1.279",
                               "model.attention.use_entropy=True"], 2, 3, 64, 64, "eval"),
    "cfg3_rgb_audio_mha_T13": (["data.flow.enable=False", "model.attention.use_entropy=True"], 1, 3, 64, 64, "eval"),
    "cfg4_all_noattn": (["model.attention.enable=False", "data.audio.audio_length=1.279", "data.sampling=async"],
                        2, 3, 64, 64, "eval"),
    "cfg5_all_mha_eval": (["data.audio.audio_length=1.279"], 2, 5, 64, 64, "eval"),
    "fixed_attn": (["data.flow.enable=False", "model.attention.use_fixed=True", "data.audio.audio_length=1.279"],
                   2, 3, 64, 64, "eval"),
    "unimodal_attn": (["data.flow.enable=False", "model.attention.type=unimodal",
                       "data.audio.audio_length=1.279"], 2, 3, 64, 64, "eval"),
    "proto_attn": (["data.flow.enable=False", "model.attention.type=proto", "data.audio.audio_length=1.279"],
                   2, 3, 64, 64, "eval"),
    # training mode: BN batch statistics, dropout disabled so the run is deterministic
    "train_cfg4_all_noattn": (["model.attention.enable=False", "data.audio.audio_length=1.279",
                               "model.fusion_dropout=0"], 2, 3, 64, 64, "train"),
    "train_cfg3_mha": (["data.flow.enable=False", "data.audio.audio_length=1.279", "model.fusion_dropout=0",
                        "model.attention.attn_dropout=0.0", "model.attention.use_entropy=True",
                        "model.attention.use_prior=True", "model.attention.use_contrast=True",
                        "model.freeze_base=False"], 2, 3, 64, 64, "train"),
}
GRAD_KEYS = ["Base_RGB.conv1_7x7_s2.weight", "Base_RGB.conv1_7x7_s2_bn.weight", "Base_RGB.inception_4a_3x3.weight",
             "Base_RGB.inception_3c_double_3x3_2.bias", "Base_Audio.conv1_7x7_s2.weight",
             "Base_Audio.inception_5b_pool_proj.weight", "Base_Flow.conv1_7x7_s2.bias",
             "Base_Audio.inception_3a_1x1_bn.weight",
             "fusion.fusion_layer.0.bias", "classifier.verb.weight", "classifier.noun.bias",
             "attention_layer.attention_layer.in_proj_bias", "attention_layer.attention_layer.out_proj.bias",
             "pe.1.bias", "pe.2.weight"]


GRAD_KEYS_EP20 = ["Base_RGB.conv1_7x7_s2.weight", "attention_layer.attention_layer.in_proj_bias", "pe.2.weight"]


def make_inputs(cfg, modality, B, n, hv, ha, seed):
    g = torch.Generator().manual_seed(seed)
    T = round(cfg.data.audio.audio_length * 25 / 4)
    W = 1 + (int(cfg.data.audio.audio_length * 24000) - 1) // 120
    inp = {}
    for m in modality:
        if m == "RGB":
            inp[m] = h16(torch.rand(B, n, 3, hv, hv, generator=g) - 0.45)
        elif m == "Flow":
            inp[m] = h16(torch.rand(B, n, 10, hv, hv, generator=g) - 0.5)
        else:
            inp[m] = h16(torch.randn(B, n, 1, ha, W, generator=g) * 3 - 6)
    if cfg.model.attention.enable and cfg.model.attention.use_fixed:
        w = torch.rand(B, n, T, 1, generator=g)
        inp["weights"] = h16(w / w.sum(2, keepdim=True))
    target = {"class": {"verb": torch.randint(0, 125, (B,), generator=g),
                        "noun": torch.randint(0, 352, (B,), generator=g)}}
    if cfg.model.attention.enable and cfg.model.attention.use_prior:
        target["weights"] = torch.from_numpy(gaussian_kernel(T, 1)).float().view(1, 1, T, 1).repeat(B, n, 1, 1)
    return inp, target, T, W


def model_cases():
    import core.models as rm
    import core.models.bn_inception as rbi
    pre = pretrained_pair(7)
    rbi.torch.load = lambda f, map_location=None: {k: v.clone() for k, v in
                                                   pre["kinetics" if "kinetics" in f else "imagenet"].items()}
    from attention_based_tbn_amd.config import get_modality
    for ci, (name, (ov, B, n, hv, ha, mode)) in enumerate(CASES.items()):
        cfg = load_config(ov)
        modality = get_modality(cfg)
        torch.manual_seed(100 + ci)
        model, crit, ngpu = rm.build_model(cfg, modality, torch.device("cpu"))
        sd = fill_state_dict(model.state_dict(), 1000 + ci)
        model.load_state_dict(sd)
        keys = [[k, list(v.shape)] for k, v in model.state_dict().items()]
        trainable = [k for k, p in model.named_parameters() if p.requires_grad]
        with open(os.path.join(HERE, f"keys_{name}.json"), "w") as f:
            json.dump({"overrides": ov, "modality": modality, "keys": keys, "trainable": trainable,
                       "fill_seed": 1000 + ci}, f)
        inp, target, T, W = make_inputs(cfg, modality, B, n, hv, ha, 2000 + ci)
        arrs = {"in_" + k: v.half() for k, v in inp.items()}
        arrs.update({"tgt_" + k: v for k, v in target["class"].items()})
        model.train(mode == "train")
        if mode == "eval":
            with torch.no_grad():
                out = model({k: v.clone() for k, v in inp.items()})
            for k, v in out.items():
                arrs["out_" + k] = v
            if "weights" in out or not cfg.model.attention.enable or cfg.model.attention.use_fixed:
                for ep in (0, 20):
                    loss, bs = model.get_loss(crit, target, out, epoch=ep)
                    for k, v in loss.items():
                        arrs[f"loss_ep{ep}_{k}"] = torch.as_tensor(v).float()
        else:
            for ep in (0, 20):
                model.zero_grad()
                # fresh running stats each time so both epochs see the same state
                model.load_state_dict(sd)
                out = model({k: v.clone() for k, v in inp.items()})
                loss, bs = model.get_loss(crit, target, out, epoch=ep)
                loss["total"].backward()
                for k, v in out.items():
                    arrs[f"ep{ep}_out_" + k] = v
                for k, v in loss.items():
                    arrs[f"ep{ep}_loss_{k}"] = torch.as_tensor(v).float()
                params = dict(model.named_parameters())
                for k in (GRAD_KEYS if ep == 0 else GRAD_KEYS_EP20):
                    if k in params and params[k].grad is not None:
                        arrs[f"ep{ep}_grad_" + k] = params[k].grad
                gn = torch.sqrt(sum((p.grad.double() ** 2).sum() for p in model.parameters() if p.grad is not None))
                arrs[f"ep{ep}_gradnorm"] = gn.float()
            # running-stat update of one BN after exactly one training forward from `sd`
            st = model.state_dict()
            for k in ("Base_RGB.conv1_7x7_s2_bn.running_mean", "Base_RGB.inception_4a_3x3_bn.running_var",
                      "Base_RGB.inception_4a_3x3_bn.num_batches_tracked"):
                arrs["post_" + k] = st[k]
        save(f"model_{name}.npz", **arrs)


# ---------------------------------------------------------------------------------------------
# Row a11 of SURVEY section 8: the two host-side rules of reference core/models/model.py
#   :215-222  audio dropout (inverted test on a host np.random.uniform() draw; the backbone still runs)
#   :243-248  crop repeat (visual rows = k x audio rows -> audio feature tiled k times, n *= k)
A11_DROPOUT = (["model.attention.enable=False", "data.audio.audio_length=1.279", "model.fusion_dropout=0",
                "data.audio.dropout=0.5"], 2, 3, 64, 64)
A11_REPEAT = (["data.flow.enable=False", "model.attention.enable=False", "data.audio.audio_length=1.279"],
              2, 3, 64, 64, 2)
A11_GRAD_KEYS = ["Base_RGB.conv1_7x7_s2.weight", "Base_Audio.conv1_7x7_s2.weight", "Base_Flow.inception_4a_3x3.weight",
                 "fusion.fusion_layer.0.bias", "classifier.verb.weight"]


def a11_cases():
    import core.models as rm
    import core.models.bn_inception as rbi
    from attention_based_tbn_amd.config import get_modality
    pre = pretrained_pair(7)
    rbi.torch.load = lambda f, map_location=None: {k: v.clone() for k, v in
                                                   pre["kinetics" if "kinetics" in f else "imagenet"].items()}
    # ---- audio dropout: one NumPy seed per branch of the inverted test (draw > dropout -> feature zeroed)
    ov, B, n, hv, ha = A11_DROPOUT
    cfg = load_config(ov)
    modality = get_modality(cfg)
    seeds = {}
    for s in range(100):
        u = np.random.RandomState(s).uniform()
        seeds.setdefault("drop" if u > cfg.data.audio.dropout else "keep", s)
        if len(seeds) == 2:
            break
    torch.manual_seed(300)
    model, crit, _ = rm.build_model(cfg, modality, torch.device("cpu"))
    sd = fill_state_dict(model.state_dict(), 3000)
    model.load_state_dict(sd)
    with open(os.path.join(HERE, "keys_train_audio_dropout.json"), "w") as f:
        json.dump({"overrides": ov, "modality": modality, "keys": [[k, list(v.shape)] for k, v in sd.items()],
                   "trainable": [k for k, p in model.named_parameters() if p.requires_grad], "fill_seed": 3000,
                   "np_seeds": seeds}, f)
    inp, target, T, W = make_inputs(cfg, modality, B, n, hv, ha, 3100)
    arrs = {"in_" + k: v.half() for k, v in inp.items()}
    arrs.update({"tgt_" + k: v for k, v in target["class"].items()})
    model.train()
    for branch, s in seeds.items():
        model.load_state_dict(sd)
        model.zero_grad()
        np.random.seed(s)
        out = model({k: v.clone() for k, v in inp.items()})
        loss, bs = model.get_loss(crit, target, out, epoch=0)
        loss["total"].backward()
        for k, v in out.items():
            arrs[f"{branch}_out_" + k] = v
        for k, v in loss.items():
            arrs[f"{branch}_loss_{k}"] = torch.as_tensor(v).float()
        params = dict(model.named_parameters())
        for k in A11_GRAD_KEYS:
            arrs[f"{branch}_hasgrad_" + k] = np.bool_(params[k].grad is not None)
            if params[k].grad is not None:
                arrs[f"{branch}_grad_" + k] = params[k].grad
        # the audio backbone runs in both branches: its BN running statistics advance either way
        arrs[f"{branch}_post_Base_Audio.conv1_7x7_s2_bn.running_mean"] = model.state_dict()[
            "Base_Audio.conv1_7x7_s2_bn.running_mean"]
    save("model_train_audio_dropout.npz", **arrs)

    # ---- crop repeat: RGB carries k x the audio rows (the reference's 10-crop testing), eval mode
    ov, B, n, hv, ha, k = A11_REPEAT
    cfg = load_config(ov)
    modality = get_modality(cfg)
    torch.manual_seed(301)
    model, crit, _ = rm.build_model(cfg, modality, torch.device("cpu"))
    sd = fill_state_dict(model.state_dict(), 3001)
    model.load_state_dict(sd)
    with open(os.path.join(HERE, "keys_crop_repeat_eval.json"), "w") as f:
        json.dump({"overrides": ov, "modality": modality, "keys": [[kk, list(v.shape)] for kk, v in sd.items()],
                   "trainable": [kk for kk, p in model.named_parameters() if p.requires_grad], "fill_seed": 3001,
                   "repeat": k}, f)
    inp, target, T, W = make_inputs(cfg, modality, B, n, hv, ha, 3101)
    g = torch.Generator().manual_seed(3102)
    inp["RGB"] = h16(torch.rand(B, n * k, 3, hv, hv, generator=g) - 0.45)      # k crops per segment
    arrs = {"in_" + kk: v.half() for kk, v in inp.items()}
    arrs.update({"tgt_" + kk: v for kk, v in target["class"].items()})
    model.eval()
    with torch.no_grad():
        out = model({kk: v.clone() for kk, v in inp.items()})
    for kk, v in out.items():
        arrs["out_" + kk] = v
    loss, bs = model.get_loss(crit, target, out, epoch=0)
    for kk, v in loss.items():
        arrs[f"loss_ep0_{kk}"] = torch.as_tensor(v).float()
    save("model_crop_repeat_eval.npz", **arrs)


# ---------------------------------------------------------------------------------------------
def sampler_cases():
    import pandas as pd
    import core.dataset.dataset as rds
    ann = pd.read_csv(REF + "/annotations/epic_train_val.csv")
    nf = ann.stop_frame - ann.start_frame
    # a spread of rows: shortest segments (edge cases seg_len<=0), medians, longest, plus random picks
    order = nf.sort_values().index
    rows = list(order[:12]) + list(order[len(order) // 2: len(order) // 2 + 6]) + list(order[-6:])
    rows += list(np.random.RandomState(0).choice(len(ann), 24, replace=False))
    rows = [int(r) for r in rows]
    sub = ann.iloc[rows].reset_index(drop=True)
    # synthetic very short segments: exercise seg_len <= 0 (all indices = start) and tiny seg_len
    extra = sub.iloc[:6].copy()
    extra["start_frame"] = [100, 100, 1, 57, 2001, 10]
    extra["stop_frame"] = [103, 108, 20, 60, 2013, 36]
    sub = pd.concat([sub, extra]).reset_index(drop=True)
    out = {"rows": [{"start_frame": int(r.start_frame), "stop_frame": int(r.stop_frame)} for r in sub.itertuples()],
           "cases": []}
    for mods in (["RGB", "Flow", "Audio"], ["RGB", "Audio"], ["Flow", "Audio"], ["Audio"]):
        for sampling in ("sync", "async"):
            for mode, nseg in (("train", 3), ("val", 25), ("test", 25), ("train", 8)):
                ds = object.__new__(rds.Video_Dataset)
                ds.cfg = load_config([f"data.sampling={sampling}"])
                ds.mode, ds.modality, ds.num_segments = mode, mods, nseg
                ds.frame_len = {m: (5 if m == "Flow" else 1) for m in mods}
                ds.read_flow_pickle, ds.use_attention = False, False
                ds.annotations = sub
                ds._get_frames = lambda m, vid, idx: ([], [])
                ds._transform_data = lambda x, m: x
                seed = 1234
                np.random.seed(seed)
                res = []
                for i in range(len(sub)):
                    item = ds[i]
                    res.append({m: [int(v) for v in item[0]["indices"][m]] for m in mods})
                out["cases"].append({"modality": mods, "sampling": sampling, "mode": mode, "num_segments": nseg,
                                     "seed": seed, "indices": res})
    with open(os.path.join(HERE, "sampler.json"), "w") as f:
        json.dump(out, f)
    print("wrote sampler.json", len(out["cases"]), "cases x", len(sub), "rows")


if __name__ == "__main__":
    install_stubs()
    torch.set_num_threads(8)
    only = sys.argv[1:]          # e.g. `make_golden.py a11` regenerates just that group
    for name, fn in (("trunk", trunk_pin), ("factory", factory_audio), ("models", model_cases), ("a11", a11_cases),
                     ("sampler", sampler_cases)):
        if not only or name in only:
            fn()
