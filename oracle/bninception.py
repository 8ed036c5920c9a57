"""Oracle (TEST INFRASTRUCTURE): BN-Inception backbone on torch-CPU, fp32.

Restates the graph the reference instantiates through the third-party
`pretrainedmodels.models.bninception.BNInception` class
(reference `core/models/bn_inception.py:5-6,11,74,90`) -- that package is not
vendored, so the layer list is taken from the in-repo statement of the same
graph, `core/models/bn_inception_audio.py:24-404` (layers) and `:437-1003`
(dataflow / concat order), with the 7x7 stem the reference really uses
(`core/models/bn_inception.py:75-77`, `bn_inception_audio.py:35-39`).

`logits()` follows the reference override `core/models/bn_inception.py:16-35`.
`bninception()` follows the factory `core/models/bn_inception.py:38-107`
except that weights come from a caller-supplied state dict (the Drive-hosted
`.pth` files are not available offline).
"""
import torch
import torch.nn as nn
import torch.nn.functional as F

# (name, cin, c1x1, c3x3_reduce, c3x3, cdbl_reduce, cdbl_1, cdbl_2, pool, cproj, stride)
# pool: "avg" = AvgPool 3x3 s1 p1 (count_include_pad) -> pool_proj 1x1
#       "max" = MaxPool 3x3 s1 p1 -> pool_proj 1x1            (inception_5b)
#       "pass" = MaxPool 3x3 s2 ceil, concatenated directly    (inception_3c / 4e)
BLOCKS = [
    ("3a", 192, 64, 64, 64, 64, 96, 96, "avg", 32, 1),
    ("3b", 256, 64, 64, 96, 64, 96, 96, "avg", 64, 1),
    ("3c", 320, 0, 128, 160, 64, 96, 96, "pass", 0, 2),
    ("4a", 576, 224, 64, 96, 96, 128, 128, "avg", 128, 1),
    ("4b", 576, 192, 96, 128, 96, 128, 128, "avg", 128, 1),
    ("4c", 576, 160, 128, 160, 128, 160, 160, "avg", 128, 1),
    ("4d", 608, 96, 128, 192, 160, 192, 192, "avg", 128, 1),
    ("4e", 608, 0, 128, 192, 192, 256, 256, "pass", 0, 2),
    ("5a", 1056, 352, 192, 320, 160, 224, 224, "avg", 128, 1),
    ("5b", 1024, 352, 192, 320, 192, 224, 224, "max", 128, 1),
]


class BNInception(nn.Module):
    """Same constructor signature / attribute names as the third-party class."""

    def __init__(self, num_classes=1000, in_channels=3):
        super().__init__()
        self._cbr("conv1_7x7_s2", "conv1_relu_7x7", in_channels, 64, 7, 2, 3)
        self.pool1_3x3_s2 = nn.MaxPool2d((3, 3), stride=(2, 2), dilation=(1, 1), ceil_mode=True)
        self._cbr("conv2_3x3_reduce", "conv2_relu_3x3_reduce", 64, 64, 1, 1, 0)
        self._cbr("conv2_3x3", "conv2_relu_3x3", 64, 192, 3, 1, 1)
        self.pool2_3x3_s2 = nn.MaxPool2d((3, 3), stride=(2, 2), dilation=(1, 1), ceil_mode=True)
        for (b, cin, c1, c3r, c3, cdr, cd1, cd2, pool, cp, st) in BLOCKS:
            p = "inception_" + b
            if c1:
                self._cbr(p + "_1x1", p + "_relu_1x1", cin, c1, 1, 1, 0)
            self._cbr(p + "_3x3_reduce", p + "_relu_3x3_reduce", cin, c3r, 1, 1, 0)
            self._cbr(p + "_3x3", p + "_relu_3x3", c3r, c3, 3, st, 1)
            self._cbr(p + "_double_3x3_reduce", p + "_relu_double_3x3_reduce", cin, cdr, 1, 1, 0)
            self._cbr(p + "_double_3x3_1", p + "_relu_double_3x3_1", cdr, cd1, 3, 1, 1)
            self._cbr(p + "_double_3x3_2", p + "_relu_double_3x3_2", cd1, cd2, 3, st, 1)
            if pool == "avg":
                setattr(self, p + "_pool", nn.AvgPool2d(3, stride=1, padding=1, ceil_mode=True,
                                                       count_include_pad=True))
            elif pool == "max":
                setattr(self, p + "_pool", nn.MaxPool2d((3, 3), stride=(1, 1), padding=(1, 1),
                                                       dilation=(1, 1), ceil_mode=True))
            else:
                setattr(self, p + "_pool", nn.MaxPool2d((3, 3), stride=(2, 2), dilation=(1, 1),
                                                       ceil_mode=True))
            if cp:
                self._cbr(p + "_pool_proj", p + "_relu_pool_proj", cin, cp, 1, 1, 0)
        self.global_pool = nn.AvgPool2d(7, stride=1, padding=0, ceil_mode=True, count_include_pad=True)
        self.last_linear = nn.Linear(1024, num_classes)
        # attributes the reference factory sets (bn_inception.py:97-99)
        self.is_audio = False
        self.attend = False
        self.feature_size = 1024

    def _cbr(self, name, relu_name, cin, cout, k, s, p):
        setattr(self, name, nn.Conv2d(cin, cout, kernel_size=(k, k), stride=(s, s), padding=(p, p)))
        setattr(self, name + "_bn", nn.BatchNorm2d(cout, affine=True))
        setattr(self, relu_name, nn.ReLU(True))

    def _run(self, name, relu_name, x):
        return getattr(self, relu_name)(getattr(self, name + "_bn")(getattr(self, name)(x)))

    def stem(self, x):
        x = self._run("conv1_7x7_s2", "conv1_relu_7x7", x)
        return self.pool1_3x3_s2(x)

    def trunk(self, x, taps=None):
        """Everything after pool1 (identical to bn_inception_audio.py:446-1003)."""
        x = self._run("conv2_3x3_reduce", "conv2_relu_3x3_reduce", x)
        x = self._run("conv2_3x3", "conv2_relu_3x3", x)
        x = self.pool2_3x3_s2(x)
        if taps is not None:
            taps["pool2_3x3_s2"] = x
        for (b, cin, c1, c3r, c3, cdr, cd1, cd2, pool, cp, st) in BLOCKS:
            p = "inception_" + b
            outs = []
            if c1:
                outs.append(self._run(p + "_1x1", p + "_relu_1x1", x))
            y = self._run(p + "_3x3_reduce", p + "_relu_3x3_reduce", x)
            outs.append(self._run(p + "_3x3", p + "_relu_3x3", y))
            y = self._run(p + "_double_3x3_reduce", p + "_relu_double_3x3_reduce", x)
            y = self._run(p + "_double_3x3_1", p + "_relu_double_3x3_1", y)
            outs.append(self._run(p + "_double_3x3_2", p + "_relu_double_3x3_2", y))
            y = getattr(self, p + "_pool")(x)
            if cp:
                y = self._run(p + "_pool_proj", p + "_relu_pool_proj", y)
            outs.append(y)
            x = torch.cat(outs, 1)
            if taps is not None:
                taps[p + "_output"] = x
        return x

    def features(self, x, taps=None):
        x = self.stem(x)
        if taps is not None:
            taps["pool1_3x3_s2"] = x
        return self.trunk(x, taps)

    def logits(self, features):
        hw = features.shape[2:]
        if self.is_audio and self.attend:
            return F.avg_pool2d(features, kernel_size=(hw[0], 1), stride=(hw[0], 1))
        x = F.avg_pool2d(features, kernel_size=hw)
        return x.view(x.size(0), -1)

    def forward(self, x):
        return self.logits(self.features(x))


def bninception(in_channels, modality, data_dict, is_audio=False, attend=False, num_classes=1000):
    """Factory following reference bn_inception.py:38-107 with an explicit state dict.

    `data_dict` plays the role of the `.pth` file: it holds a 3-channel
    (`imagenet`) or `in_channels`-channel (`kinetics`, Flow) first conv.
    """
    data_dict = dict(data_dict)
    model = BNInception(num_classes=num_classes)
    if modality == "Audio":
        model.conv1_7x7_s2 = nn.Conv2d(in_channels, 64, kernel_size=(7, 7), stride=(2, 2), padding=(3, 3))
        data_dict["conv1_7x7_s2.weight"] = data_dict["conv1_7x7_s2.weight"].mean(dim=1).unsqueeze(dim=1)
        sd = model.state_dict()
        for k in [k for k in sd.keys() if k not in data_dict]:
            data_dict[k] = sd[k]
        for k in [k for k in data_dict.keys() if k not in sd]:
            del data_dict[k]
    elif modality == "Flow":
        model.conv1_7x7_s2 = nn.Conv2d(in_channels, 64, kernel_size=(7, 7), stride=(2, 2), padding=(3, 3))
    model.is_audio = is_audio
    model.attend = attend
    model.feature_size = 1024
    model.load_state_dict(data_dict)
    delattr(model, "last_linear")
    return model


def seeded_state_dict(seed, in_channels=3, num_classes=1000):
    """Deterministic stand-in for the pretrained `.pth` files (random but well scaled).

    BN running stats and affine terms are randomised so that eval-mode BN is not
    a near-identity (which would hide bugs).
    """
    g = torch.Generator().manual_seed(seed)
    ref = BNInception(num_classes=num_classes, in_channels=in_channels)
    sd = {}
    for k, v in ref.state_dict().items():
        if k.endswith("num_batches_tracked"):
            sd[k] = torch.zeros((), dtype=torch.long)
        elif k.endswith("running_var"):
            sd[k] = 0.5 + torch.rand(v.shape, generator=g)
        elif k.endswith("running_mean"):
            sd[k] = 0.2 * torch.randn(v.shape, generator=g)
        elif "_bn.weight" in k:
            sd[k] = 0.8 + 0.4 * torch.rand(v.shape, generator=g)
        elif "_bn.bias" in k:
            sd[k] = 0.1 * torch.randn(v.shape, generator=g)
        elif k.endswith(".bias"):
            sd[k] = 0.1 * torch.randn(v.shape, generator=g)
        else:  # conv / linear weights: He-style so activations keep O(1) scale
            fan_in = v[0].numel()
            sd[k] = torch.randn(v.shape, generator=g) * (2.0 / fan_in) ** 0.5
    return sd
