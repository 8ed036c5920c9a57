"""ORACLE (test infrastructure only -- never imported by the product path).

NumPy restatement of the reference's visual input pipeline, core/dataset/transform.py:9-543 as composed by
core/utils/create_dataloader.py:19-81, on the reference's own data type (a list of uint8 HxW(xC) arrays):

  multiscale_crop / sample_crop_size / fill_fix_offset   transform.py:284-413
  rescale                                                transform.py:222-281
  center_crop, random_flip                               transform.py:60-103, 182-219
  stack, to_tensor, normalize                            transform.py:415-543

`resize_linear_u8` restates cv2.resize(..., interpolation=cv2.INTER_LINEAR) for 8-bit images from OpenCV's published
algorithm (modules/imgproc/src/resize.cpp: fixed-point weights with INTER_RESIZE_COEF_BITS = 11, cvRound, integer
horizontal pass, vertical pass ((b0*(S0>>4))>>16 + (b1*(S1>>4))>>16 + 2) >> 2).  cv2 (opencv-python, un-pinned in
the reference's install/requirements.txt) is NOT installed in the build image: **that function is parity unpinned**.
Everything else is pinned by tests/golden/transform.npz, produced by the unmodified reference classes with
`cv2.resize` stubbed by this function (tests/golden/make_golden_trainstep.py).
"""
import numpy as np


def _coefs(dsize, ssize, clamp_frac):
    scale = 1.0 / (float(dsize) / float(ssize))
    d = np.arange(dsize, dtype=np.float64)
    f = ((d + 0.5) * scale - 0.5).astype(np.float32)
    s = np.floor(f).astype(np.int64)
    f = (f - s.astype(np.float32)).astype(np.float32)
    if clamp_frac:
        lo, hi = s < 0, s >= ssize - 1
        f[lo], s[lo] = 0.0, 0
        f[hi], s[hi] = 0.0, ssize - 1
    a0 = np.rint((np.float32(1.0) - f) * np.float32(2048.0)).astype(np.int64)   # cvRound: half to even
    a1 = np.rint(f * np.float32(2048.0)).astype(np.int64)
    return s, a0, a1


def resize_linear_u8(img, new_w, new_h):
    """cv2.resize(img, (new_w, new_h), interpolation=cv2.INTER_LINEAR) for uint8 HxW or HxWxC"""
    img = np.asarray(img)
    squeeze = img.ndim == 2
    src = img.reshape(img.shape[0], img.shape[1], -1).astype(np.int64)
    h, w = src.shape[:2]
    sx, ax0, ax1 = _coefs(new_w, w, True)
    sy, ay0, ay1 = _coefs(new_h, h, False)
    x1 = np.minimum(sx + 1, w - 1)
    hor = src[:, sx, :] * ax0[None, :, None] + src[:, x1, :] * ax1[None, :, None]       # (h, new_w, c)
    y0, y1 = np.clip(sy, 0, h - 1), np.clip(sy + 1, 0, h - 1)
    v = (((ay0[:, None, None] * (hor[y0] >> 4)) >> 16) + ((ay1[:, None, None] * (hor[y1] >> 4)) >> 16) + 2) >> 2
    out = np.clip(v, 0, 255).astype(np.uint8)
    return out[:, :, 0] if squeeze else out


def fill_fix_offset(more_fix_crop, image_w, image_h, crop_w, crop_h):
    w_step, h_step = (image_w - crop_w) / 4, (image_h - crop_h) / 4
    ret = [(0, 0), (4 * w_step, 0), (0, 4 * h_step), (4 * w_step, 4 * h_step), (2 * w_step, 2 * h_step)]
    if more_fix_crop:
        ret += [(0, 2 * h_step), (4 * w_step, 2 * h_step), (2 * w_step, 4 * h_step), (2 * w_step, 0 * h_step),
                (1 * w_step, 1 * h_step), (3 * w_step, 1 * h_step), (1 * w_step, 3 * h_step), (3 * w_step, 3 * h_step)]
    return ret


def sample_crop_size(im_size, input_size, scales, max_distort=1, more_fix_crop=True):
    """two np.random.randint draws, as the reference (fix_crop=True)"""
    img_h, img_w = im_size
    base = min(img_w, img_h)
    sizes = [int(base * x) for x in scales]
    crop_h = [input_size[1] if abs(x - input_size[1]) < 3 else x for x in sizes]
    crop_w = [input_size[0] if abs(x - input_size[0]) < 3 else x for x in sizes]
    pairs = [(w, h) for i, h in enumerate(crop_h) for j, w in enumerate(crop_w) if abs(i - j) <= max_distort]
    cw, ch = pairs[np.random.randint(len(pairs))]
    offs = fill_fix_offset(more_fix_crop, img_w, img_h, cw, ch)
    ow, oh = offs[np.random.randint(len(offs))]
    return cw, ch, int(ow), int(oh)


def rescale(imgs, size):
    h, w = imgs[0].shape[:2]
    if isinstance(size, int):
        new_h, new_w = (size * h / w, size) if h > w else (size, size * w / h)
    else:
        new_h, new_w = size
    new_h, new_w = int(new_h), int(new_w)
    return [im if (new_h, new_w) == im.shape[:2] else resize_linear_u8(im, new_w, new_h) for im in imgs]


def train_geometry(imgs, input_size, scales):
    """MultiScaleCrop -> RandomHorizontalFlip"""
    cw, ch, ow, oh = sample_crop_size(imgs[0].shape[:2], (input_size, input_size), scales)
    imgs = rescale([im[oh:oh + ch, ow:ow + cw] for im in imgs], (input_size, input_size))
    if np.random.random() < 0.5:
        imgs = [np.fliplr(im) for im in imgs]
    return imgs


def test_geometry(imgs, scale_size, crop_size):
    """Rescale -> CenterCrop"""
    imgs = rescale(imgs, scale_size)
    out = []
    for im in imgs:
        x1, y1 = (im.shape[1] - crop_size) // 2, (im.shape[0] - crop_size) // 2
        out.append(im[y1:y1 + crop_size, x1:x1 + crop_size])
    return out


def stack_totensor_normalize(imgs, modality, mean, std, length=10):
    """Stack -> ToTensor -> Normalize: float32 (n, C, h, w)"""
    c = 3 if modality == "RGB" else 1
    arr = [np.asarray(im).reshape(1, im.shape[0], im.shape[1], c)[0] for im in imgs]
    if modality == "Flow":
        arr = [np.concatenate(arr[i:i + length], axis=2) for i in range(0, len(arr), length)]
    x = np.stack(arr, 0).transpose(0, 3, 1, 2).astype(np.float32)
    x = x / np.float32(255)
    ch = x.shape[1]
    m = np.resize(np.asarray(mean, dtype=np.float32), ch) if len(mean) < ch else np.asarray(mean, dtype=np.float32)
    s = np.resize(np.asarray(std, dtype=np.float32), ch) if len(std) < ch else np.asarray(std, dtype=np.float32)
    return ((x - m.reshape(1, -1, 1, 1)) / s.reshape(1, -1, 1, 1)).astype(np.float32)
