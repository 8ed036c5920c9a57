"""Visual input pipeline.
CPU: the oracle against the fixture produced by the unmodified reference pipelines (cv2.resize stubbed -- the
interpolation arithmetic itself is parity-unpinned, see oracle/transform.py); known-answer checks of the restated
cv2 INTER_LINEAR.  GPU: the fused HIP pipeline bit-exact against the same fixture and the oracle."""
import json
import os

import numpy as np
import pytest
import torch

from oracle import transform as otf

GOLD = os.path.join(os.path.dirname(__file__), "golden")


def _cases():
    meta = json.load(open(os.path.join(GOLD, "transform.json")))
    z = np.load(os.path.join(GOLD, "transform.npz"))
    return meta, z


def _oracle_run(cfg, case, frames):
    m = case["modality"]
    node = cfg.data.rgb if m == "RGB" else cfg.data.flow
    np.random.seed(case["seed"])
    if case["mode"] == "train":
        sc = [1, 0.875, 0.75, 0.66] if m == "RGB" else [1, 0.875, 0.75]
        imgs = otf.train_geometry(list(frames), cfg.data.train_crop_size, sc)
    else:
        imgs = otf.test_geometry(list(frames), cfg.data.test_scale_size, cfg.data.test_crop_size)
    return otf.stack_totensor_normalize(imgs, m, list(node.mean), list(node.std))


def test_oracle_pipeline_matches_reference_fixture():
    from attention_based_tbn_amd.config import load_config
    meta, z = _cases()
    cfg = load_config(meta["overrides"])
    modes = set()
    for k, case in enumerate(meta["cases"]):
        got = _oracle_run(cfg, case, z["in%d" % k])
        assert list(got.shape) == case["shape"]
        assert np.array_equal(got, z["out%d" % k]), case
        modes.add((case["mode"], case["modality"]))
    assert len(modes) == 4


def test_restated_cv2_linear_resize_known_answers():
    rng = np.random.RandomState(0)
    img = rng.randint(0, 256, (37, 53, 3)).astype(np.uint8)
    assert np.array_equal(otf.resize_linear_u8(img, 53, 37), img)                      # identity
    flat = np.full((20, 30), 77, np.uint8)
    assert np.array_equal(otf.resize_linear_u8(flat, 45, 31), np.full((31, 45), 77, np.uint8))   # constants survive
    # exact 2x upsampling of a horizontal ramp: interior samples sit at 1/4, 3/4 between neighbours
    ramp = np.tile((np.arange(16) * 8).astype(np.uint8), (4, 1))
    up = otf.resize_linear_u8(ramp, 32, 4).astype(int)
    assert up[0, 0] == 0 and up[0, -1] == 120
    assert np.array_equal(up[0, 1:-1:2], ramp[0, :-1].astype(int) + 2) and np.array_equal(up[0, 2:-1:2], ramp[0, 1:].astype(int) - 2)
    # halving averages 2x2 blocks (weights 1/2, 1/2), rounded
    a = rng.randint(0, 256, (8, 8)).astype(np.uint8)
    half = otf.resize_linear_u8(a, 4, 4).astype(int)
    blk = a.reshape(4, 2, 4, 2).astype(int).sum((1, 3))
    assert np.all(np.abs(half - blk / 4.0) <= 1.0)
    # monotone input stays within the source range
    out = otf.resize_linear_u8(img, 224, 224)
    assert out.dtype == np.uint8 and out.shape == (224, 224, 3)


def test_restated_cv2_resize_agrees_with_float_bilinear():
    """independent check of the sampling geometry: cv2's INTER_LINEAR is half-pixel-centred bilinear without
    antialiasing -- torch's `interpolate(mode="bilinear", align_corners=False, antialias=False)` -- evaluated in
    11-bit fixed point; the 8-bit results must sit within one grey level of the float evaluation, up- and down-scaling"""
    import torch
    import torch.nn.functional as F
    rng = np.random.RandomState(5)
    for (h, w), (nh, nw) in (((256, 456), (224, 399)), ((120, 160), (256, 341)), ((37, 53), (224, 224)), ((300, 200), (75, 50))):
        img = rng.randint(0, 256, (h, w, 3)).astype(np.uint8)
        got = otf.resize_linear_u8(img, nw, nh).astype(np.float64)
        ref = F.interpolate(torch.from_numpy(img).permute(2, 0, 1)[None].double(), size=(nh, nw), mode="bilinear",
                            align_corners=False, antialias=False)[0].permute(1, 2, 0).numpy()
        assert np.abs(got - ref).max() <= 1.0, ((h, w), (nh, nw), np.abs(got - ref).max())
        assert abs((got - ref).mean()) < 0.25          # the truncating shifts of the fixed-point form bias by ~0.1 level


def test_host_draw_order_matches_reference_sampler():
    """the recorded geometry of the device pipeline consumes the NumPy RNG exactly like the oracle / reference"""
    from attention_based_tbn_amd.core.dataset.transform import MultiScaleCrop, RandomHorizontalFlip, _Geometry
    for seed in range(20):
        np.random.seed(seed)
        geo = _Geometry(57, 45)
        MultiScaleCrop(32, [1, 0.875, 0.75, 0.66])(geo)
        RandomHorizontalFlip(0.5)(geo)
        after = np.random.random()
        np.random.seed(seed)
        cw, ch, ow, oh = otf.sample_crop_size((57, 45), (32, 32), [1, 0.875, 0.75, 0.66])
        flip = np.random.random() < 0.5
        assert geo.box == [ow, oh, cw, ch] and geo.flip == flip and after == np.random.random()
        assert geo.resized == (None if (cw, ch) == (32, 32) else (32, 32))


@pytest.mark.gpu
def test_device_pipeline_matches_reference_fixture_bit_exact():
    from attention_based_tbn_amd.config import load_config
    from attention_based_tbn_amd.core.dataset import get_transforms
    meta, z = _cases()
    cfg = load_config(meta["overrides"])
    tfs = {mode: get_transforms(cfg, ["RGB", "Flow", "Audio"], mode) for mode in ("train", "test")}
    for k, case in enumerate(meta["cases"]):
        frames = [f for f in z["in%d" % k]]
        np.random.seed(case["seed"])
        got = tfs[case["mode"]][case["modality"]](frames)
        assert got.is_cuda and list(got.shape) == case["shape"]
        assert np.array_equal(got.cpu().numpy(), z["out%d" % k]), case


@pytest.mark.gpu
@pytest.mark.parametrize("hw,mode", [((256, 456), "train"), ((256, 456), "test"), ((240, 320), "test"), ((300, 200), "train")])
def test_device_pipeline_full_size_vs_oracle(hw, mode):
    """EPIC-sized frames at the real crop sizes (224 / 256): HIP kernel == oracle, bit for bit"""
    from attention_based_tbn_amd.config import load_config
    from attention_based_tbn_amd.core.dataset import get_transforms
    cfg = load_config([])
    rng = np.random.RandomState(hw[0] + len(mode))
    tfs = get_transforms(cfg, ["RGB", "Flow"], mode)
    for m, n_img in (("RGB", 3), ("Flow", 30)):
        frames = [rng.randint(0, 256, hw + ((3,) if m == "RGB" else ())).astype(np.uint8) for _ in range(n_img)]
        for seed in (1, 2, 3):
            np.random.seed(seed)
            got = tfs[m](frames)
            want = _oracle_run(cfg, {"modality": m, "mode": mode, "seed": seed}, frames)
            assert np.array_equal(got.cpu().numpy(), want), (m, seed)
    # a uint8 tensor already on the device is accepted as well
    t = torch.from_numpy(np.stack(frames[:10], 0)).reshape(10, hw[0], hw[1], 1).cuda()
    np.random.seed(5)
    a = tfs["Flow"](t)
    np.random.seed(5)
    b = tfs["Flow"](frames[:10])
    assert torch.equal(a, b)


@pytest.mark.gpu
def test_device_pipeline_rejects_bad_input():
    from attention_based_tbn_amd._lib import TbnHipError
    from attention_based_tbn_amd.config import load_config
    from attention_based_tbn_amd.core.dataset import get_transforms
    tf = get_transforms(load_config([]), ["RGB", "Flow"], "test")
    with pytest.raises(TbnHipError):
        tf["RGB"](torch.zeros(2, 256, 300, 3))                       # float, not uint8
    with pytest.raises(TbnHipError):
        tf["Flow"]([np.zeros((256, 300), np.uint8)] * 7)             # not a multiple of the stack length
