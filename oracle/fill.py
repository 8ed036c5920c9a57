"""Oracle (TEST INFRASTRUCTURE): deterministic, name-keyed parameter fill.

Both the reference model (in tests/golden/make_golden.py) and the oracle / product
models (in tests) are filled by `fill_state_dict`, which derives every tensor from
a hash of its state-dict *key* and its shape only -- so two models get identical
weights exactly when their key names and shapes agree, independent of constructor
order.  Values are scaled so activations stay O(1) through the network.
"""
import zlib

import torch


def _gen(key, seed):
    return torch.Generator().manual_seed((zlib.crc32(key.encode()) ^ (seed * 2654435761)) & 0x7FFFFFFF)


def fill_tensor(key, ref, seed):
    g = _gen(key, seed)
    shape = tuple(ref.shape)
    if key.endswith("num_batches_tracked"):
        return torch.zeros(shape, dtype=torch.long)
    if key.endswith("running_var"):
        return 0.5 + torch.rand(shape, generator=g)
    if key.endswith("running_mean"):
        return 0.2 * torch.randn(shape, generator=g)
    if key.endswith("pe.0.pe") or key.endswith("prototype_wts"):
        return ref.clone()  # deterministic buffers keep their constructor value
    is_norm = "_bn." in key or key.endswith("pe.2.weight") or key.endswith("pe.2.bias")
    if is_norm and key.endswith("weight"):
        return 0.8 + 0.4 * torch.rand(shape, generator=g)
    if key.endswith("bias"):
        return 0.1 * torch.randn(shape, generator=g)
    if len(shape) >= 2:  # conv / linear / in_proj weights
        fan_in = 1
        for d in shape[1:]:
            fan_in *= d
        gain = 2.0 if len(shape) == 4 else 1.0
        return torch.randn(shape, generator=g) * (gain / fan_in) ** 0.5
    return torch.randn(shape, generator=g)


def fill_state_dict(sd, seed):
    """Returns a new dict with the same keys/shapes, values derived from (key, seed)."""
    return {k: fill_tensor(k, v, seed) for k, v in sd.items()}


def pretrained_pair(seed):
    """Stand-ins for weights/imagenet_bninception_rgb.pth and kinetics_bninception_flow.pth."""
    from .bninception import BNInception
    rgb = fill_state_dict(BNInception(1000, 3).state_dict(), seed)
    flow = fill_state_dict(BNInception(400, 10).state_dict(), seed + 1)
    return {"imagenet": rgb, "kinetics": flow}
