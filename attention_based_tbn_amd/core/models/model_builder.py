"""`build_model(cfg, modality, device) -> (model, criterion, num_gpus)` -- mirror of reference
core/models/model_builder.py:25-81.  Multi-GPU: instead of wrapping in nn.DataParallel when more
than one GPU is visible to ONE process, the model is wrapped in the RCCL `DataParallel` when the
process is one rank of an initialised torch.distributed job (torchrun, one process per GPU)."""
from collections import OrderedDict

import torch
import torch.distributed as dist

from .contrast_loss import ContrastLoss
from .dataparallel import DataParallel
from .model import TBNModel

_MODEL_TYPES = {"bninception": TBNModel}

_LOSS_TYPES = {
    "crossentropy": torch.nn.CrossEntropyLoss,
    "nll": torch.nn.NLLLoss,
    "kl": torch.nn.KLDivLoss,
    "mse": torch.nn.MSELoss,
    "smoothl1": torch.nn.SmoothL1Loss,
}


def build_model(cfg, modality, device, pretrained_state=None):
    assert cfg.model.arch in _MODEL_TYPES.keys(), \
        "Model type '{}' not supported on the MI355X hot path (bninception only)".format(cfg.model.arch)
    assert cfg.model.loss_fn in _LOSS_TYPES.keys(), "Loss type '{}' not supported".format(cfg.model.loss_fn)
    world = dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1
    requested = len(cfg.gpu_ids) if len(cfg.gpu_ids) > 0 else (max(world, 1) if device.type == "cuda" else 0)
    if device.type == "cuda" and requested > 1 and world == 1:
        # the reference would wrap in single-process nn.DataParallel here (model_builder.py:73-75); this path is one
        # process per GPU -- fail now, not at the first checkpoint an epoch later
        raise RuntimeError(f"gpu_ids lists {requested} GPUs but this process is not a rank of a torch.distributed job: "
                           f"launch one process per GPU (python -m torch.distributed.run --nproc-per-node {requested} ...)")
    assert world == 1 or len(cfg.gpu_ids) in (0, world), \
        f"gpu_ids lists {len(cfg.gpu_ids)} GPUs but the torch.distributed world has {world} ranks"
    # > 1 exactly when the model is wrapped (callers unwrap `.module` on it, reference misc.py:93-112)
    num_gpus = world if (world > 1 and device.type == "cuda") else min(requested, 1)

    model = _MODEL_TYPES[cfg.model.arch](cfg, modality, device, pretrained_state=pretrained_state)

    criterion = OrderedDict()
    criterion[cfg.model.loss_fn] = _LOSS_TYPES[cfg.model.loss_fn]()
    if cfg.model.attention.enable:
        if cfg.model.attention.use_prior:
            criterion["prior"] = _LOSS_TYPES[cfg.model.attention.wt_loss](reduction=cfg.model.attention.loss_reduction)
        if cfg.model.attention.use_contrast:
            criterion["contrast"] = ContrastLoss(threshold=cfg.model.attention.contrast_thresh,
                                                 reduction=cfg.model.attention.loss_reduction)

    model = model.to(device)
    if world > 1 and device.type == "cuda":
        model = DataParallel(model)
    for key in criterion.keys():
        criterion[key] = criterion[key].to(device)
    return model, criterion, num_gpus
