"""Visual input pipeline on the MI355X (reference `core/dataset/transform.py:9-543`, composed by
`core/utils/create_dataloader.py:19-81`).

The reference transforms a Python list of uint8 HxWxC frames on the host, image by image (crop, `cv2.resize`, flip,
stack, `/255`, normalise) and ships fp32 tensors to the GPU.  Here every transform only *records* its decision
(same constructor arguments, same NumPy RNG draws in the same order, so a seeded run picks the same crop box and
flip); the pixels move to the device as uint8 (4x fewer PCIe bytes) and ONE HIP kernel (`tbn_frames_to_tensor`)
produces the normalised fp32 NCHW tensor.  `get_transforms(cfg, modality, mode)` returns the same dictionary of
callables as the reference.  No CPU fallback: the pipeline raises without a GPU.
"""
from collections import OrderedDict

import numpy as np
import torch

from ..._lib import TbnHipError, call, ptr, stream_ptr


class _Geometry:
    """what the recorded transforms do to a frame of (h, w): source box -> resized size -> crop window -> flip"""

    def __init__(self, h, w):
        self.h, self.w = h, w           # current logical size
        self.box = [0, 0, w, h]         # x, y, w, h in the SOURCE frame (only valid before a resize)
        self.resized = None             # (w, h) once a resize happened
        self.crop = None                # x, y, w, h inside the resized box
        self.flip = False

    def do_crop(self, x, y, w, h):
        if self.resized is None:
            self.box = [self.box[0] + x, self.box[1] + y, w, h]
        else:
            c = self.crop or [0, 0, self.resized[0], self.resized[1]]
            self.crop = [c[0] + x, c[1] + y, w, h]
        self.w, self.h = w, h

    def do_resize(self, new_w, new_h):
        if self.resized is not None:
            raise TbnHipError("input pipeline: at most one resize per pipeline (as in the reference's compositions)")
        if (new_h, new_w) != (self.h, self.w):
            self.resized = (new_w, new_h)
            self.w, self.h = new_w, new_h


class MultiScaleCrop(object):
    """reference transform.py:284-413 -- same constructor, same two `np.random.randint` draws"""

    def __init__(self, input_size, scales=[1, 0.875, 0.75, 0.66], max_distort=1, fix_crop=True, more_fix_crop=True):
        self.scales, self.max_distort, self.fix_crop, self.more_fix_crop = scales, max_distort, fix_crop, more_fix_crop
        assert isinstance(input_size, (int, tuple))
        self.input_size = input_size if isinstance(input_size, tuple) else (input_size, input_size)

    def __call__(self, geo):
        crop_w, crop_h, off_w, off_h = self._sample_crop_size((geo.h, geo.w))
        geo.do_crop(off_w, off_h, crop_w, crop_h)
        Rescale(self.input_size)(geo)
        return geo

    def _sample_crop_size(self, im_size):
        img_h, img_w = im_size[0], im_size[1]
        base_size = min(img_w, img_h)
        crop_sizes = [int(base_size * x) for x in self.scales]
        crop_h = [self.input_size[1] if abs(x - self.input_size[1]) < 3 else x for x in crop_sizes]
        crop_w = [self.input_size[0] if abs(x - self.input_size[0]) < 3 else x for x in crop_sizes]
        pairs = [(w, h) for i, h in enumerate(crop_h) for j, w in enumerate(crop_w) if abs(i - j) <= self.max_distort]
        crop_pair = pairs[np.random.randint(len(pairs))]
        if not self.fix_crop:
            w_offset = np.random.randint(0, img_w - crop_pair[0])
            h_offset = np.random.randint(0, img_h - crop_pair[1])
        else:
            offsets = self.fill_fix_offset(self.more_fix_crop, img_w, img_h, crop_pair[0], crop_pair[1])
            w_offset, h_offset = offsets[np.random.randint(len(offsets))]
        return crop_pair[0], crop_pair[1], int(w_offset), int(h_offset)

    @staticmethod
    def fill_fix_offset(more_fix_crop, image_w, image_h, crop_w, crop_h):
        w_step, h_step = (image_w - crop_w) / 4, (image_h - crop_h) / 4
        ret = [(0, 0), (4 * w_step, 0), (0, 4 * h_step), (4 * w_step, 4 * h_step), (2 * w_step, 2 * h_step)]
        if more_fix_crop:
            ret += [(0, 2 * h_step), (4 * w_step, 2 * h_step), (2 * w_step, 4 * h_step), (2 * w_step, 0 * h_step),
                    (1 * w_step, 1 * h_step), (3 * w_step, 1 * h_step), (1 * w_step, 3 * h_step),
                    (3 * w_step, 3 * h_step)]
        return ret


class Rescale(object):
    """reference transform.py:222-281 (size: int = smaller edge, or (h, w))"""

    def __init__(self, size, interpolation=1):
        assert isinstance(size, (int, tuple))
        self.size = size

    def __call__(self, geo):
        h, w = geo.h, geo.w
        if isinstance(self.size, int):
            new_h, new_w = (self.size * h / w, self.size) if h > w else (self.size, self.size * w / h)
        else:
            new_h, new_w = self.size
        geo.do_resize(int(new_w), int(new_h))
        return geo


class CenterCrop(object):
    """reference transform.py:60-103"""

    def __init__(self, size):
        self.size = (size, size) if isinstance(size, int) else size

    def __call__(self, geo):
        h, w = self.size
        geo.do_crop((geo.w - w) // 2, (geo.h - h) // 2, w, h)
        return geo


class RandomCrop(object):
    """reference transform.py:9-57"""

    def __init__(self, size):
        self.size = (size, size) if isinstance(size, int) else size

    def __call__(self, geo):
        th, tw = self.size
        x1 = np.random.randint(0, geo.w - tw)
        y1 = np.random.randint(0, geo.h - th)
        if not (geo.w == tw and geo.h == th):
            geo.do_crop(x1, y1, tw, th)
        return geo


class RandomHorizontalFlip(object):
    """reference transform.py:182-219 -- one `np.random.random()` draw per sample"""

    def __init__(self, prob=0.5):
        self.prob = prob

    def __call__(self, geo):
        if np.random.random() < self.prob:
            geo.flip = not geo.flip
        return geo


class DevicePipeline(object):
    """A composed visual pipeline: geometry transforms (recorded) + Stack + ToTensor + Normalize (executed by one
    kernel).  Call with the reference's argument -- a list of uint8 HxW(xC) arrays of one sample -- or with a
    uint8 tensor (n_img, H, W, C); returns the fp32 (n_img / stack, C * stack, h, w) tensor on the device."""

    def __init__(self, modality, geometry, mean, std, length=10, device="cuda"):
        self.modality, self.geometry = modality, list(geometry)
        self.stack = length if modality == "Flow" else 1
        self.channels = 3 if modality == "RGB" else 1
        self.device = torch.device(device)
        self.mean = torch.tensor(mean, dtype=torch.float32)
        self.std = torch.tensor(std, dtype=torch.float32)
        self._stat_dev = None

    def _stats(self):
        if self._stat_dev is None:
            self._stat_dev = (self.mean.to(self.device), self.std.to(self.device))
        return self._stat_dev

    def __call__(self, img_list):
        if not torch.cuda.is_available():
            raise TbnHipError("input pipeline: needs an MI355X (no CPU fallback)")
        if isinstance(img_list, torch.Tensor):
            frames = img_list
        else:
            assert isinstance(img_list, list) and len(img_list) > 0
            arr = np.stack([np.asarray(im).reshape(im.shape[0], im.shape[1], self.channels) for im in img_list], 0)
            frames = torch.from_numpy(np.ascontiguousarray(arr))
        if frames.dtype != torch.uint8 or frames.dim() != 4 or frames.shape[3] != self.channels:
            raise TbnHipError(f"input pipeline: expected uint8 frames (n, H, W, {self.channels}), got "
                              f"{frames.dtype} {tuple(frames.shape)}")
        frames = frames.to(self.device, non_blocking=True).contiguous()
        n, H, W, C = frames.shape
        if n % self.stack != 0:
            raise TbnHipError(f"input pipeline: {n} frames is not a multiple of the stack length {self.stack}")
        geo = _Geometry(H, W)
        for t in self.geometry:
            t(geo)
        rw, rh = geo.resized if geo.resized is not None else (geo.box[2], geo.box[3])
        cx, cy, ow, oh = geo.crop if geo.crop is not None else (0, 0, rw, rh)
        mean, std = self._stats()
        out = torch.empty((n // self.stack, C * self.stack, oh, ow), dtype=torch.float32, device=self.device)
        call("tbn_frames_to_tensor", ptr(frames), n, H, W, C, geo.box[0], geo.box[1], geo.box[2], geo.box[3], rw, rh,
             cx, cy, ow, oh, int(geo.flip), self.stack, ptr(mean), ptr(std), mean.numel(), 1, ptr(out), stream_ptr())
        return out


class AudioToTensor(object):
    """Stack + ToTensor(is_audio=True) of the reference for spectrograms: list of (256, W) arrays -> (n, 1, 256, W)"""

    def __init__(self, device="cuda"):
        self.device = torch.device(device)

    def __call__(self, img_list):
        arr = np.stack([np.asarray(im).reshape(im.shape[0], im.shape[1], 1) for im in img_list], 0)
        return torch.from_numpy(arr).permute(0, 3, 1, 2).contiguous().float().to(self.device, non_blocking=True)


def get_transforms(cfg, modality, mode="test", device="cuda"):
    """reference create_dataloader.py:19-81: the same compositions per modality and mode"""
    transforms = OrderedDict()
    for m in modality:
        if m in ("RGB", "Flow"):
            node = cfg.data.rgb if m == "RGB" else cfg.data.flow
            if mode == "train":
                scales = [1, 0.875, 0.75, 0.66] if m == "RGB" else [1, 0.875, 0.75]
                geometry = [MultiScaleCrop(cfg.data.train_crop_size, scales), RandomHorizontalFlip(prob=0.5)]
            else:
                geometry = [Rescale(cfg.data.test_scale_size), CenterCrop(cfg.data.test_crop_size)]
            transforms[m] = DevicePipeline(m, geometry, list(node.mean), list(node.std), device=device)
        elif m == "Audio":
            transforms[m] = AudioToTensor(device)
    return transforms
