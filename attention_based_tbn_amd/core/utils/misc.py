"""Checkpoint and score-file formats of the reference (`core/utils/misc.py:56-165`), unchanged on the wire:
`torch.save` of {epoch, train_loss, validation_loss, validation_accuracy, optimizer, model[, conf_mat, scheduler]}
with the reference's `state_dict` key names (the flat GPU parameter storage is unpacked by the model), and the
EPIC-Kitchens `action_recognition` JSON of `save_scores`."""
import json
import os

import numpy as np
import torch

from ...config import get_modality  # noqa: F401  (reference exports it from misc as well)


def get_time_diff(start_time, end_time):
    hours = int((end_time - start_time) / 3600)
    minutes = int((end_time - start_time) / 60) - (hours * 60)
    seconds = int(np.floor((end_time - start_time) % 60))
    return (hours, minutes, seconds)


def save_checkpoint(model, optimizer, epoch, train_loss_hist, val_loss_hist, val_acc_hist, confusion_matrix,
                    num_gpus, scheduler=None, filename="checkpoint.pth"):
    target = getattr(model, "module", model) if num_gpus > 1 else model
    # the optimiser state is written over the reference's per-layer parameter list, so the file loads in the
    # reference (`optimizer.load_state_dict`) as well as here
    data = {"epoch": epoch, "train_loss": train_loss_hist, "validation_loss": val_loss_hist,
            "validation_accuracy": val_acc_hist, "optimizer": optimizer_state_to_reference(target, optimizer)}
    data["model"] = target.state_dict()
    if confusion_matrix:
        data["conf_mat"] = confusion_matrix
    if scheduler:
        data["scheduler"] = scheduler.state_dict()
    torch.save(data, filename)


def load_checkpoint(filename, model, optimizer=None, scheduler=None, num_gpus=1, map_location="cpu"):
    """Inverse of `save_checkpoint` (the reference inlines this in train.py:219-240 / test.py:60-66).  Accepts
    checkpoints written by the reference: their per-layer momentum buffers are scattered into the flat ones."""
    data = torch.load(filename, map_location=map_location)
    target = model.module if (num_gpus > 1 and hasattr(model, "module")) else model
    target.load_state_dict(data["model"])
    if optimizer is not None and "optimizer" in data:
        n_saved = sum(len(g["params"]) for g in data["optimizer"]["param_groups"])
        n_here = sum(len(g["params"]) for g in optimizer.param_groups)
        if n_saved == n_here:
            optimizer.load_state_dict(data["optimizer"])
        else:   # written by the reference: one slot per per-layer parameter
            optimizer_state_from_reference(target, optimizer, data["optimizer"])
    if scheduler is not None and "scheduler" in data:
        scheduler.load_state_dict(data["scheduler"])
    return data


FLAT_NAMES = ("flat_weight", "flat_bias", "bn_weight_first", "bn_weight_rest", "bn_bias_first", "bn_bias_rest")


def _backbones(model):
    from ..models.bn_inception import BNInception
    return [(n, m) for n, m in model.named_modules() if isinstance(m, BNInception)]


def reference_parameter_names(model):
    """Names of `model.parameters()` of the REFERENCE model, in its order (= the index space of a reference
    checkpoint's optimizer state): the state_dict keys minus the buffers."""
    own = dict(model.named_parameters())
    bb = _backbones(model)
    names = []
    for key in model.state_dict().keys():
        pre = next((n for n, _ in bb if key.startswith(n + ".") and not key.startswith(n + ".last_linear.")), None)
        if pre is not None:
            if key.endswith((".running_mean", ".running_var", ".num_batches_tracked")):
                continue
            names.append(key)
        elif key in own:
            names.append(key)
    return names


def optimizer_state_from_reference(model, optimizer, ref_state):
    """Load a reference checkpoint's SGD state (one momentum buffer per per-layer parameter) into an optimiser
    built over THIS model's parameters (flat backbone storage): buffers are scattered into flat ones."""
    names = reference_parameter_names(model)
    ref_idx = [i for g in ref_state["param_groups"] for i in g["params"]]
    assert len(ref_idx) == len(names), f"reference optimizer has {len(ref_idx)} parameters, the model {len(names)}"
    by_name = {n: ref_state["state"].get(i, {}).get("momentum_buffer") for n, i in zip(names, ref_idx)}
    own = {p: n for n, p in model.named_parameters()}
    bb = dict(_backbones(model))
    done = {}
    for pre, m in bb.items():
        flat = {a: torch.zeros_like(getattr(m, a)) for a in FLAT_NAMES}
        owner = {flat[a].untyped_storage().data_ptr(): a for a in FLAT_NAMES}
        filled = set()
        for key, view in m.reference_param_views(flat):
            buf = by_name.get(pre + "." + key)
            if buf is not None:
                view.copy_(buf.to(view.device))
                filled.add(owner[view.untyped_storage().data_ptr()])
        for a in FLAT_NAMES:   # a flat tensor none of whose layers had a buffer (frozen BN affine) keeps none
            done[getattr(m, a)] = flat[a] if a in filled else None
    for group in optimizer.param_groups:
        for p in group["params"]:
            buf = done[p] if p in done else by_name.get(own.get(p))
            if buf is not None:
                optimizer.state[p]["momentum_buffer"] = buf.to(p.device).clone()
    for k in ("lr", "momentum", "weight_decay"):
        if k in ref_state["param_groups"][0]:
            for g in optimizer.param_groups:
                g[k] = ref_state["param_groups"][0][k]


def optimizer_state_to_reference(model, optimizer):
    """`optimizer.state_dict()` re-expressed over the reference's per-layer parameter list (what the reference's
    `optim.SGD.load_state_dict` expects)."""
    names = reference_parameter_names(model)
    own = dict(model.named_parameters())
    per_name = {}
    for pre, m in _backbones(model):
        flat = {}
        for a in FLAT_NAMES:
            st = optimizer.state.get(getattr(m, a), {})
            flat[a] = st.get("momentum_buffer")
        if all(v is None for v in flat.values()):
            continue
        have = {a for a, v in flat.items() if v is not None}
        flat = {a: (v if v is not None else torch.zeros_like(getattr(m, a))) for a, v in flat.items()}
        owner = {flat[a].untyped_storage().data_ptr(): a for a in FLAT_NAMES}
        for key, view in m.reference_param_views(flat):
            if owner[view.untyped_storage().data_ptr()] in have:   # frozen tensors have no buffer, as in torch
                per_name[pre + "." + key] = view.detach().clone(memory_format=torch.contiguous_format)
    state = {}
    for i, n in enumerate(names):
        if n in per_name:
            state[i] = {"momentum_buffer": per_name[n]}
        elif n in own and optimizer.state.get(own[n], {}).get("momentum_buffer") is not None:
            state[i] = {"momentum_buffer": optimizer.state[own[n]]["momentum_buffer"].detach().clone()}
    g0 = {k: v for k, v in optimizer.param_groups[0].items() if k != "params"}
    g0["params"] = list(range(len(names)))
    return {"state": state, "param_groups": [g0]}


def save_scores(scores, file_name, action_names):
    """reference misc.py:115-165, same JSON document"""
    out_result = {"version": "0.1", "challenge": "action_recognition"}
    for key in scores.keys():
        scores[key] = torch.cat(scores[key], dim=0)
    results = {}
    ids = scores["action_id"].cpu()
    host = {k: v.detach().float().cpu() for k, v in scores.items() if k != "action_id"}
    for idx in range(ids.shape[0]):
        a_id = str(ids[idx].item())
        results[a_id] = {}
        for key, val in host.items():
            if key == "action":
                top, ind = val[idx].topk(100, dim=0, largest=True, sorted=True)   # only the top 100 action scores
                results[a_id][key] = {action_names[i.item()]: s.item() for i, s in zip(ind, top)}
            else:
                results[a_id][key] = {str(i): s.item() for i, s in enumerate(val[idx])}
    out_result["results"] = results
    os.makedirs(os.path.split(file_name)[0], exist_ok=True)
    with open(file_name, "w") as f:
        json.dump(out_result, f, indent=4)
