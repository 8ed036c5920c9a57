"""Hydra-compatible config tree for the TBN hot path (no hydra / omegaconf dependency).

Mirrors the reference config surface: `config/config.yaml:1-12` (defaults list +
top-level keys) and the groups `config/{data,model,train,val,test}/tbn_*.yaml`
(keys + defaults documented in `config/CONFIG.md`).  `load_config(overrides)`
returns a tree with attribute access (`cfg.model.attention.enable`), dict access
(`cfg.model.num_classes.keys()`) and `cfg.pretty()` like the OmegaConf object the
reference's `main.py:17-28` receives; overrides use the Hydra CLI syntax
`group.key=value`.  `load_config(config_dir=...)` composes a user-supplied
Hydra-style directory (a `config.yaml` with a `defaults:` list) instead of the
built-in defaults, so the reference's own `config/` directory can be used as is.
"""
import copy
import os

import re

import yaml


class _Loader(yaml.SafeLoader):
    """SafeLoader that also reads `1e-2` as a float (as OmegaConf's loader does)."""


_Loader.add_implicit_resolver(
    "tag:yaml.org,2002:float",
    re.compile(r"""^(?:[-+]?(?:[0-9][0-9_]*)\.[0-9_]*(?:[eE][-+]?[0-9]+)?
                    |[-+]?(?:[0-9][0-9_]*)(?:[eE][-+]?[0-9]+)
                    |\.[0-9_]+(?:[eE][-+][0-9]+)?
                    |[-+]?\.(?:inf|Inf|INF)|\.(?:nan|NaN|NAN))$""", re.X),
    list("-+0123456789."))


def _yload(text):
    return yaml.load(text, Loader=_Loader)


class ConfigNode(dict):
    """dict with attribute access, recursively."""

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError:
            raise AttributeError(k)

    def __setattr__(self, k, v):
        self[k] = _wrap(v)

    def pretty(self):
        return yaml.safe_dump(_unwrap(self), default_flow_style=False, sort_keys=False)

    def __deepcopy__(self, memo):
        return _wrap(copy.deepcopy(_unwrap(self), memo))


def _wrap(v):
    if isinstance(v, dict) and not isinstance(v, ConfigNode):
        return ConfigNode({k: _wrap(x) for k, x in v.items()})
    if isinstance(v, list):
        return [_wrap(x) for x in v]
    return v


def _unwrap(v):
    if isinstance(v, dict):
        return {k: _unwrap(x) for k, x in v.items()}
    if isinstance(v, list):
        return [_unwrap(x) for x in v]
    return v


def _defaults():
    rgb = dict(enable=True, dir_prefix="links", file_ext="jpg", mean=[0.408, 0.459, 0.502], std=[1.0, 1.0, 1.0])
    flow = dict(enable=True, read_flow_pickle=False, dir_prefix="links", file_ext="jpg", win_length=5,
                mean=[0.502], std=[1.0])
    audio = dict(enable=True, read_audio_pickle=False, dir_prefix="audio", sampling_rate=24000,
                 audio_length=2.1, spec_type="stft", file_ext="wav", dropout=0)
    data = dict(dataset="epic", sampling="sync", rgb=rgb, flow=flow, audio=audio, vid_fps=60,
                train_scale_size=256, train_crop_size=224, test_scale_size=256, test_crop_size=224,
                manual_seed=0)
    attention = dict(enable=True, use_pe=True, type="mha", use_gumbel=True, use_fixed=False,
                     prior_type="gaussian", attn_heads=4, attn_dropout=0.5, use_prior=False, wt_loss="kl",
                     wt_decay=0.25, loss_reduction="batchmean", use_contrast=False, contrast_thresh=0.1,
                     contrast_decay=0.25, use_entropy=False, entropy_decay=0.25, entropy_thresh=0.2,
                     decay_step=10)
    model = dict(arch="bninception", attention=attention, resnet=dict(depth=101), vgg=dict(type="16"),
                 freeze_base=True, freeze_mode="partialbn", num_classes=dict(verb=125, noun=352),
                 agg_type="avg", fusion_dropout=0.5, loss_fn="crossentropy", checkpoint_dir="tbn_weights")
    train = dict(enable=True, annotation_file="annotations/epic_train_val.csv",
                 vid_list="data/train_split_seen.txt", batch_size=12, epochs=30,
                 optim=dict(type="sgd", lr=1e-2, momentum=0.9, weight_decay=0, accumulator_step=1),
                 scheduler=dict(lr_steps=[20], lr_decay=1e-1),
                 warmup=dict(enable=False, multiplier=1, epochs=5),
                 clip_grad=20, num_segments=3, pre_trained="")
    val = dict(enable=True, vid_list="data/val_split_seen.txt", batch_size=2, topk=[1, 5], num_segments=25,
               pre_trained="")
    test = dict(enable=False,
                annotation_file=["annotations/EPIC_test_s1_timestamps.csv",
                                 "annotations/EPIC_test_s2_timestamps.csv"],
                vid_list="", batch_size=2, topk=[1, 5], num_segments=25, save_results=False,
                results_file=["seen.json", "unseen.json"], pre_trained="")
    return dict(data=data, model=model, train=train, val=val, test=test, num_workers=8, gpu_ids=[],
                exp_name="attention_test/seen/", data_dir="/media/data/tridiv/epic",
                out_dir="/media/data/tridiv/epic")


def _merge(dst, src):
    for k, v in src.items():
        if isinstance(v, dict) and isinstance(dst.get(k), dict):
            _merge(dst[k], v)
        else:
            dst[k] = v


def _compose_dir(config_dir):
    with open(os.path.join(config_dir, "config.yaml")) as f:
        root = _yload(f.read()) or {}
    tree = {}
    for item in root.pop("defaults", []):
        (group, name), = item.items()
        if group == "hydra":
            continue
        with open(os.path.join(config_dir, group, name + ".yaml")) as f:
            _merge(tree, _yload(f.read()) or {})
    _merge(tree, root)
    return tree


def apply_overrides(tree, overrides):
    for ov in overrides or []:
        if "=" not in ov:
            raise ValueError(f"override '{ov}' is not of the form key.path=value")
        path, raw = ov.split("=", 1)
        val = _yload(raw)
        node = tree
        keys = path.split(".")
        for k in keys[:-1]:
            if k not in node or not isinstance(node[k], dict):
                node[k] = {}
            node = node[k]
        node[keys[-1]] = val
    return tree


def load_config(overrides=None, config_dir=None):
    tree = _compose_dir(config_dir) if config_dir else _defaults()
    apply_overrides(tree, overrides)
    return _wrap(tree)


def get_modality(cfg):
    """reference core/utils/misc.py:7-26 -- fixed RGB, Flow, Audio order."""
    out = []
    if cfg.data.rgb.enable:
        out.append("RGB")
    if cfg.data.flow.enable:
        out.append("Flow")
    if cfg.data.audio.enable:
        out.append("Audio")
    return out
