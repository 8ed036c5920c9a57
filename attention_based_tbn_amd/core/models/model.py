"""Temporal Binding Network with attention-weighted mid-level fusion on the MI355X HIP path.

API mirror of reference core/models/model.py: `TBNModel(cfg, modality, device)` (:21-101),
`forward(input: dict) -> OrderedDict` (:205-262), `get_loss(criterion, target, preds, epoch)`
(:264-334), `Fusion` (:337-362), `Classifier` (:365-386); same parameter names, so reference
checkpoints load.  Differences are implementation only: each backbone is one engine call, the
modality features are produced by HIP pooling kernels, fusion / classifier / attention
projections run on the fp32-MFMA GEMM, and the per-class classifiers run as ONE GEMM.
"""
import os
import weakref
from collections import OrderedDict

import numpy as np
import torch
import torch.nn as nn
from torch.distributions import Categorical

from ... import ops
from .attention import MultiheadedAttention, PositionalEncoding, PrototypeAttention, UniModalAttention
from .bn_inception import bninception


class _ArenaCatFn(torch.autograd.Function):
    """the concatenation of features that ALREADY sit side by side in `arena`: no copy forward, column views backward"""

    @staticmethod
    def forward(ctx, arena, *feats):
        ctx.widths = [f.shape[1] for f in feats]
        return arena.detach()

    @staticmethod
    def backward(ctx, g):
        out, o = [], 0
        for w in ctx.widths:
            out.append(g[:, o:o + w])
            o += w
        return (None,) + tuple(out)


def _concat_features(arena, features):
    """torch.cat(features, dim=1) (reference model.py:250) -- free when every feature is its column range of `arena`"""
    if arena is not None and sum(f.shape[1] for f in features) == arena.shape[1]:
        o, ok = 0, True
        for f in features:
            ok = ok and (f.dim() == 2 and f.shape[0] == arena.shape[0] and f.stride() == (arena.shape[1], 1)
                         and f.data_ptr() == arena.data_ptr() + 4 * o and f.dtype == arena.dtype)
            o += f.shape[1]
        if ok:
            return _ArenaCatFn.apply(arena, *features)
    return torch.cat(features, dim=1)


class _PadColsFn(torch.autograd.Function):
    """Conv1d(k=1) weight (N, K, 1) as a GEMM operand (N, kp) with K zero-padded to kp columns; cached until the parameter
    changes (its version counter), like ops.cat_pad_rows"""

    @staticmethod
    def forward(ctx, cache, kp, weight):
        key = (weight.data_ptr(), weight._version, kp)
        if cache.get("key") != key:
            buf = cache.get("buf")
            if buf is None or tuple(buf.shape) != (weight.shape[0], kp) or buf.device != weight.device:
                buf = cache["buf"] = torch.zeros(weight.shape[0], kp, device=weight.device, dtype=weight.dtype)
            buf[:, :weight.shape[1]].copy_(weight.detach()[:, :, 0])
            cache["key"] = key
        ctx.k = weight.shape[1]
        return cache["buf"].detach()     # a fresh tensor object per call (shares storage and version counter with the cache)

    @staticmethod
    def backward(ctx, g):
        return None, None, g[:, :ctx.k].unsqueeze(2)


class _PEStack(nn.Sequential):
    """`self.pe` of the reference (model.py:62-67): PositionalEncoding -> Conv1d(1034,1024,1) -> GroupNorm(64,1024)
    with the reference's child names (pe.0.pe, pe.1.weight, pe.1.bias, pe.2.weight, pe.2.bias)."""

    def forward_sequence(self, seq):
        """native (R, T, 1024) -> (R, T, 1024)"""
        conv, gn = self[1], self[2]
        R, T, C = seq.shape
        kin = conv.weight.shape[1]
        kp = (kin + 31) // 32 * 32
        x = self[0].forward_sequence(seq, kp)
        w = _PadColsFn.apply(self.__dict__.setdefault("_wcache", {}), kp, conv.weight)   # (1024, 1034) -> x32 columns, cached
        y = ops.linear(x.view(R * T, kp), w, conv.bias)
        return ops.group_norm(y.view(R, T, -1), gn.weight, gn.bias, gn.num_groups, gn.eps)

    def forward(self, x):
        """reference layout (R, 1024, 1, T) -> (R, 1024, T)"""
        return self.forward_sequence(x.squeeze(2).transpose(1, 2).contiguous()).transpose(1, 2)


class TBNModel(nn.Module):
    IN_CHANNELS = {"RGB": 3, "Flow": 10, "Audio": 1}

    def __init__(self, cfg, modality, device, pretrained_state=None):
        """`pretrained_state`: optional {"imagenet": sd, "kinetics": sd} replacing the `.pth` files
        under <repo>/weights (reference model.py:122-125); None + missing files -> random init."""
        super().__init__()
        self.cfg = cfg
        self.modality = modality
        self.base_model_name = cfg.model.arch
        self.num_classes = cfg.model.num_classes
        self.use_attention = cfg.model.attention.enable
        self.attention_type = cfg.model.attention.type
        self.device = device
        self._pretrained_state = pretrained_state
        # MI355X: the per-modality backbones are independent until the fusion -> run them on
        # separate HIP streams so small-grid layers of one backbone fill CUs another leaves idle
        # (autograd replays each backward on its forward stream, so backward overlaps as well)
        self.multi_stream = True
        self._streams = {}
        # the heaviest backbone's stream gets HIP priority -1 (profiles/r04_ab_stream_priority.txt: Audio 36.03 ms against
        # Audio+Flow 36.14 / Flow 36.56 / none 36.38).  Plain attributes, set before the first forward: A/B runs change them
        # from their own scripts (bench.py --high-prio / --share-stream); the product reads no environment variable
        self.high_priority_modalities = ("Audio",)
        # the backbones' data-gradient weight copies (0.1 ms each, the first launch of a backward pass) are issued right after
        # the modality streams are joined for the heads, so they run beside the heads' small launches instead of in the
        # serial turn between forward and backward (BNInception.flip_weights_early; bench.py --no-early-flip for the A/B)
        self.flip_weights_early = True
        self.shared_streams = {}          # {"Flow": "RGB"}: Flow's backbone runs on RGB's stream (fewer concurrent chains)
        if cfg.model.agg_type.lower() == "avg":
            self.agg_type = "avg"
        else:
            print("Incorrect aggregation type")
            self.agg_type = None

        in_features = 0
        for m in self.modality:
            self.add_module("Base_{}".format(m), self._create_base_model(m))
            in_features += getattr(self, "Base_{}".format(m)).feature_size
            # a second (weight-gradient) stream inside a backbone pays only while it is the sole backbone
            # (measured: +7 % with one modality, 0 / -3 % once the modality streams already fill the GPU)
            getattr(self, "Base_{}".format(m)).use_aux_stream = len(self.modality) == 1
            # likewise the branch-level side stream inside a backbone (include/tbn_hip.h, tbn_backbone_params.side_stream)
            getattr(self, "Base_{}".format(m)).use_branch_streams = len(self.modality) == 1
            # with several modality streams the weight gradients of conv2_3x3 / conv2_3x3_reduce are issued AFTER conv1's
            # pooled BN backward (TBN_BACKBONE_STEM_WGRAD_LAST: same kernels, bit-identical): every backward pass then ends
            # on GEMMs, and its HBM-bound stem kernels sit under the other streams' GEMMs instead of beside their stem
            # kernels -- the three passes end together (round 6, same-box alternations: config 4 -0.06 ... -0.18 ms, 7 of 8;
            # config 3 -0.16 ms; profiles/r06_ab_stem_wgrad_last.txt)
            getattr(self, "Base_{}".format(m)).stem_wgrad_last = len(self.modality) > 1
            if cfg.model.freeze_base:
                self._freeze_base_model(m, freeze_mode=cfg.model.freeze_mode)

        if len(self.modality) > 1:
            att = cfg.model.attention
            if self.use_attention and not att.use_fixed:
                attn_win_size = round(self.cfg.data.audio.audio_length * (25 / 4))
                if att.use_pe:
                    self.pe = _PEStack(PositionalEncoding(10, max_len=attn_win_size, device=device),
                                       nn.Conv1d(1034, 1024, kernel_size=1), nn.GroupNorm(64, 1024))
                if self.attention_type == "mha":
                    self.attention_layer = MultiheadedAttention(1024, att.attn_heads, att.attn_dropout)
                elif self.attention_type == "unimodal":
                    self.attention_layer = UniModalAttention(1024, attn_win_size, hidden_size=256,
                                                             use_gumbel=att.use_gumbel, temperature=1, one_hot=True)
                elif self.attention_type == "proto":
                    self.attention_layer = PrototypeAttention(1024, attn_win_size, hidden_size=256,
                                                              use_gumbel=att.use_gumbel, temperature=1, device=device)
            self.add_module("fusion", Fusion(in_features, 512, dropout=cfg.model.fusion_dropout))
            self.add_module("classifier", Classifier(self.num_classes, 512))
        else:
            self.add_module("classifier", Classifier(self.num_classes, in_features))

    def _create_base_model(self, modality):
        if self.base_model_name != "bninception":
            raise NotImplementedError("the MI355X hot path implements the BN-Inception backbone "
                                      f"(north star); arch '{self.base_model_name}' is out of scope")
        pretrained = "kinetics" if modality == "Flow" else "imagenet"
        model_dir = os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(
            os.path.dirname(os.path.abspath(__file__))))), "weights")
        state = None
        if self._pretrained_state is not None:
            state = self._pretrained_state[pretrained]
        else:
            fname = "kinetics_bninception_flow.pth" if pretrained == "kinetics" else "imagenet_bninception_rgb.pth"
            if not os.path.exists(os.path.join(model_dir, fname)):
                pretrained = None  # weights are Drive-hosted in the reference; offline -> random init
        return bninception(self.IN_CHANNELS[modality], modality, model_dir=model_dir, pretrained=pretrained,
                           is_audio=(modality == "Audio"), attend=self.use_attention, state_dict=state)

    def _freeze_base_model(self, modality, freeze_mode):
        base = getattr(self, "Base_{}".format(modality))
        if freeze_mode == "all":
            print("Freezing the Base model.")
            for param in base.parameters():
                param.requires_grad = False
        elif freeze_mode == "partialbn" and self.base_model_name == "bninception":
            print("Freezing the batchnorms of Base Model {} except first or new layers.".format(modality))
            # reference: every BatchNorm2d child with index > 1, i.e. all but conv1_7x7_s2_bn
            base.set_bn_trainable(first=True, rest=False)

    def maybe_unused_parameter_prefixes(self):
        """parameters that may get no gradient in a step: with `data.audio.dropout > 0` the per-replica host draw of
        reference model.py:215-222 can drop the audio feature on one data-parallel rank and keep it on another --
        `DataParallel` reduces these with a rank-independent schedule (zero-filled where absent)"""
        if (self.training and len(self.modality) > 1 and "Audio" in self.modality
                and self.cfg.data.audio.dropout > 0):
            return ["Base_Audio.", "pe.", "attention_layer."]
        return []

    def optional_parameters_used(self):
        """whether the last forward kept the audio feature (the host draw of reference model.py:219): lets `DataParallel`
        exchange the decision right after forward instead of discovering it at the end of backward"""
        return not self._audio_dropped

    def _aggregate_scores(self, scores, new_shape=(1, -1)):
        assert isinstance(scores, (dict, torch.Tensor))
        assert isinstance(new_shape, tuple)
        b, n = new_shape[0], new_shape[1]
        if isinstance(scores, dict):
            for key in scores.keys():
                scores[key] = ops.segment_mean(scores[key], b, n)
            return scores
        return ops.segment_mean(scores, b, n)

    def _run_backbones(self, input):
        """raw backbone outputs per modality: (R,1024), or (R,T,1024) for attended audio"""
        # mid-level fusion concatenates the modality features (reference model.py:250): every backbone writes its pooled
        # feature straight into its column range of ONE (R, 1024 M) buffer -- the torch.cat copy and the three slice copies
        # of its backward are gone (forward() hands the buffer to the fusion layer when every feature really sits in it)
        self._arena = None
        first_in = input[self.modality[0]]
        if len(self.modality) > 1 and first_in.is_cuda:
            R = first_in.shape[0] * first_in.shape[1]
            self._arena = torch.empty(R, 1024 * len(self.modality), device=first_in.device, dtype=torch.float32)
            for i, m in enumerate(self.modality):
                getattr(self, "Base_{}".format(m))._out_slot = self._arena[:, 1024 * i:1024 * (i + 1)]
        try:
            return self._run_backbones_inner(input)
        finally:
            for m in self.modality:
                getattr(self, "Base_{}".format(m))._out_slot = None

    def _run_backbones_inner(self, input):
        def run(m):
            b, n, c, h, w = input[m].shape
            base = getattr(self, "Base_{}".format(m))
            x = input[m].reshape(b * n, c, h, w)
            return base.forward_sequence(x) if (m == "Audio" and self.use_attention) else base(x)

        first = input[self.modality[0]]
        if not (self.multi_stream and first.is_cuda and len(self.modality) > 1):
            return {m: run(m) for m in self.modality}
        main = torch.cuda.current_stream()
        raw = {}
        share = self.shared_streams
        for m in self.modality:
            if m in share and share[m] in self._streams:
                self._streams[m] = self._streams[share[m]]
            st = self._streams.get(m)
            if st is None or st.device != first.device:
                # the heaviest backbone (audio: 5.1 GFLOP per 256x256 frame against 4.1 / 4.6) gets the high-priority
                # stream: it is the one that finishes last and runs alone at the end of forward and backward (same-box
                # A/B, 3 of 3: 36.13 -> 35.92 ms per step; issuing it first instead: +-0)
                st = self._streams[m] = torch.cuda.Stream(device=first.device,
                                                          priority=(-1 if m in self.high_priority_modalities else 0))
            st.wait_stream(main)
            with torch.cuda.stream(st):
                raw[m] = run(m)
        for m in self.modality:
            main.wait_stream(self._streams[m])
            raw[m].record_stream(main)
        if self.flip_weights_early and torch.is_grad_enabled():
            # behind the join: the heads (on `main`) do not wait for these launches; each backbone's backward pass, which
            # autograd replays on the same modality stream, finds its data-gradient weight copy done
            for m in self.modality:
                with torch.cuda.stream(self._streams[m]):
                    getattr(self, "Base_{}".format(m)).flip_weights_early()
        return raw

    def forward(self, input):
        features = []
        att_wts = None
        att = self.cfg.model.attention
        self._audio_dropped = False
        raw_all = self._run_backbones(input)
        for m_no, m in enumerate(self.modality):
            b, n, c, h, w = input[m].shape
            if m == "Audio":
                # the backbone always runs (BN running statistics advance even when the audio
                # feature is then dropped), exactly like reference model.py:214-222
                raw = raw_all[m]
                if (self.training and len(self.modality) > 1 and self.cfg.data.audio.dropout > 0
                        and np.random.uniform() > self.cfg.data.audio.dropout):
                    feature = torch.zeros_like(features[0])
                    self._audio_dropped = True
                elif self.use_attention:
                    seq = raw                                           # (R, T, 1024)
                    if att.use_fixed:
                        feature = ops.weighted_sum(seq, input["weights"].reshape(b * n, -1).to(seq.device))
                    elif self.attention_type == "mha":
                        seq = self.pe.forward_sequence(seq)
                        feature, att_wts = self.attention_layer.attend(features[0], seq)
                        att_wts = att_wts.unsqueeze(1)                  # (R, 1, T) like nn.MultiheadAttention
                    elif self.attention_type in ["unimodal", "proto"]:
                        feature, att_wts = self.attention_layer.attend(features[0], seq)
                else:
                    feature = raw
                if m_no > 0 and features[0].shape[0] > feature.shape[0]:
                    new_size = features[0].shape[0] // feature.shape[0]
                    feature = feature.repeat(new_size, 1)
                    n *= new_size
            else:
                feature = raw_all[m]
            features.extend([feature])
        if len(features) > 1:
            features = _concat_features(self._arena, features)
        else:
            features = features[0]
        self._arena = None

        if len(self.modality) > 1:
            features = self.fusion(features)

        out = self.classifier(features, consensus=(b, n))

        if self.use_attention and not att.use_fixed and len(self.modality) > 1:
            if att_wts is None:
                # reference model.py:259-260 reads `att_wts` although the dropped-audio branch (:215-222) never binds it
                raise UnboundLocalError("local variable 'att_wts' referenced before assignment (audio dropout with a "
                                        "trainable attention layer: the reference fails the same way)")
            out["weights"] = att_wts
        return out

    def _fused_cross_entropy(self, criterion, target, preds):
        """{key: loss} from ONE launch when the criterion is a default nn.CrossEntropyLoss and the predictions are the very
        tensors this model's classifier returned from its last forward (Classifier.shared_scores: the heads then sit side
        by side in one score matrix), else None (the criterion is then called per key exactly as the reference does,
        model.py:272-279)"""
        ce = criterion.get("crossentropy")
        keys = list(target["class"].keys())
        if (type(ce) is not nn.CrossEntropyLoss or ce.weight is not None or ce.ignore_index != -100
                or ce.reduction != "mean" or getattr(ce, "label_smoothing", 0.0) != 0.0 or not 1 <= len(keys) <= 4):
            return None
        shared = self.classifier.shared_scores(preds, keys)
        if shared is None:
            return None
        base, heads = shared
        labels = []
        for k in keys:
            lab = target["class"][k]
            if (not base.is_cuda or not torch.is_tensor(lab) or not lab.is_cuda or lab.dtype != torch.int64 or lab.dim() != 1
                    or lab.shape[0] != base.shape[0]):
                return None
            labels.append(lab)
        return dict(zip(keys, ops.cross_entropy_heads(base, heads, labels)))

    def get_loss(self, criterion, target, preds, epoch=0):
        assert isinstance(target, dict)
        assert isinstance(preds, dict)
        assert isinstance(criterion, dict)
        att = self.cfg.model.attention
        loss = {"total": 0, "all_class": 0}
        fused = self._fused_cross_entropy(criterion, target, preds) if hasattr(self, "classifier") else None
        for key in target["class"].keys():
            labels = target["class"][key]
            batch_size = target["class"][key].shape[0]
            loss[key] = fused[key] if fused is not None else criterion["crossentropy"](preds[key], labels)
            loss["all_class"] += loss[key]
        loss["total"] += loss["all_class"]

        if self.use_attention and not att.use_fixed:
            if self.training and epoch + 1 < att.decay_step:
                prior_multiplier = contrast_multiplier = entropy_multiplier = 0
            else:
                prior_multiplier = att.wt_decay
                contrast_multiplier = att.contrast_decay
                entropy_multiplier = att.entropy_decay
            wts = preds["weights"].squeeze(1)
            if att.use_prior:
                b, n, _, _ = target["weights"].shape
                assert wts.shape[0] == b * n
                prior = target["weights"].reshape(b * n, -1)
                if att.wt_loss == "kl":
                    wts = torch.log(wts + 1e-7)
                loss["prior"] = criterion["prior"](wts, prior)
                loss["total"] += prior_multiplier * loss["prior"]
            if att.use_contrast:
                loss["contrast"] = criterion["contrast"](wts)
                loss["total"] += contrast_multiplier * loss["contrast"]
            if att.use_entropy:
                # validate_args=False: torch 1.4 (the reference's, install/requirements.txt:12) does not validate distribution
                # arguments; current torch does by default, with a `.all()` the HOST reads -- a device synchronisation in
                # the middle of every step (round 5: the host ran in lock-step with the GPU in config 3, so one 80-ms Python
                # GC pause showed up as a 130-ms step)
                loss["entropy"] = Categorical(probs=wts + 1e-6, validate_args=False).entropy().mean()
                if self.training and entropy_multiplier > 0:
                    # reference model.py:327-329: `if loss["entropy"] < entropy_thresh: entropy_multiplier = 0` -- the same
                    # switch-off as a device-side factor (a Python `if` on a GPU tensor is a host synchronisation)
                    entropy_multiplier = entropy_multiplier * (loss["entropy"].detach() >= att.entropy_thresh).to(loss["entropy"].dtype)
                loss["total"] += entropy_multiplier * loss["entropy"]
        return loss, batch_size


class Fusion(nn.Module):
    """Linear(in, out) -> ReLU -> Dropout; the bias add and ReLU are the GEMM epilogue."""

    def __init__(self, in_size, out_size, dropout=0):
        super().__init__()
        self.in_size, self.out_size, self.dropout = in_size, out_size, dropout
        self.fusion_layer = nn.Sequential(nn.Linear(in_size, out_size), nn.ReLU())
        torch.nn.init.normal_(self.fusion_layer[0].weight, 0, 1e-3)
        torch.nn.init.constant_(self.fusion_layer[0].bias, 0)
        if self.dropout > 0:
            self.dropout_layer = nn.Dropout(p=self.dropout)

    def forward(self, input):
        lin = self.fusion_layer[0]
        out = ops.linear(input, lin.weight, lin.bias, relu=True)
        return ops.dropout(out, self.dropout, self.training)


class Classifier(nn.Module):
    """one nn.Linear per class key (same names as the reference) evaluated as a single GEMM"""

    def __init__(self, num_classes, in_features):
        super().__init__()
        self.num_classes = num_classes
        self._wcache, self._bcache = {}, {}
        for cls in num_classes.keys():
            self.add_module(cls, nn.Linear(in_features, self.num_classes[cls]))
            torch.nn.init.normal_(getattr(self, cls).weight, 0, 1e-3)
            torch.nn.init.constant_(getattr(self, cls).bias, 0)

    def forward(self, input, consensus=None):
        keys = list(self.num_classes.keys())
        # all heads as one GEMM operand, zero-padded to x32 rows: cached, rebuilt only when a head's parameters change
        # (round-4 verdict: two concatenations + two pad fills per step before)
        w = ops.cat_pad_rows(self._wcache, 32, *[getattr(self, k).weight for k in keys])
        b = ops.cat_pad_rows(self._bcache, 32, *[getattr(self, k).bias for k in keys])
        scores = ops.linear(input, w, b)
        if consensus is not None:                       # temporal consensus on the fused score matrix
            scores = ops.segment_mean(scores, consensus[0], consensus[1])
        out = OrderedDict()
        layout = {}
        o = 0
        for k in keys:
            # contiguous per-head tensors, as the reference's per-key nn.Linear outputs are (model.py:365-386)
            out[k] = scores[:, o:o + self.num_classes[k]].contiguous()
            layout[k] = (weakref.ref(out[k]), o, self.num_classes[k], out[k]._version)
            o += self.num_classes[k]
        # where the heads sit in the shared score matrix stays with the MODULE (round-5 verdict: it used to ride on the
        # returned tensors as an attribute, which any op on them silently dropped): TBNModel.get_loss asks shared_scores()
        self._shared = (scores, layout)
        return out

    def shared_scores(self, preds, keys):
        """(score matrix, [(first column, classes) per key]) when preds[key] are exactly the tensors the LAST forward
        returned -- same objects, not written to since -- else None.  Lets TBNModel.get_loss evaluate the cross entropy of
        all heads in one launch on the matrix the GEMM produced (ops.cross_entropy_heads), gradient included."""
        shared = getattr(self, "_shared", None)
        if shared is None:
            return None
        scores, layout = shared
        heads = []
        for k in keys:
            ent = layout.get(k)
            t = preds.get(k) if hasattr(preds, "get") else None
            if ent is None or t is None or ent[0]() is not t or t._version != ent[3] or not scores.is_contiguous():
                return None
            heads.append((ent[1], ent[2]))
        return scores, heads
