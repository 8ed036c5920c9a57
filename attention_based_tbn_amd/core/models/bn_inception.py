"""BN-Inception backbone on the HIP engine, behind the reference's module contract.

Mirrors reference `core/models/bn_inception.py`: class `BNInception` (attributes `feature_size`,
`is_audio`, `attend`; methods `features()`, `logits()`, `forward()`, :11-35) and the factory
`bninception(in_channels, modality, pretrained, model_dir, is_audio, attend)` (:38-107).

MI355X-first storage: all 69 convs + 69 BatchNorms of a backbone live in six flat tensors laid out
by the C++ engine (`tbn_backbone_conv_info`), so a training step touches a handful of large
parameter / gradient tensors (one RCCL all-reduce bucket per backbone) and the 1x1 convs sharing a
block input run as one GEMM.  `state_dict()` / `load_state_dict()` still speak the reference's
checkpoint keys (`conv1_7x7_s2.weight` OIHW, `conv1_7x7_s2_bn.running_mean`, ... in the
reference's registration order), so checkpoints written by `core/utils/misc.py:56-112` load as is.
"""
import ctypes as C
import os
import weakref
from collections import OrderedDict

import torch
import torch.nn as nn

from ... import _lib
from ..._lib import BUCKET_CB, BackboneGrads, BackboneParams, ConvInfo, TbnHipError, call, lib, ptr, stream_ptr
from ... import ops

# block table: name, has_1x1, has_pool_proj   (layer hyper-parameters come from the engine)
_BLOCKS = [("3a", True, True), ("3b", True, True), ("3c", False, False), ("4a", True, True), ("4b", True, True),
           ("4c", True, True), ("4d", True, True), ("4e", False, False), ("5a", True, True), ("5b", True, True)]


def reference_conv_order():
    """conv module names in the reference's registration order (= state_dict key order)"""
    names = ["conv1_7x7_s2", "conv2_3x3_reduce", "conv2_3x3"]
    for b, has1, hasp in _BLOCKS:
        p = "inception_" + b
        if has1:
            names.append(p + "_1x1")
        names += [p + "_3x3_reduce", p + "_3x3", p + "_double_3x3_reduce", p + "_double_3x3_1", p + "_double_3x3_2"]
        if hasp:
            names.append(p + "_pool_proj")
    return names


class _Plan:
    """engine plan + workspace pool for one (frames, H, W)"""

    def __init__(self, cin, frames, h, w):
        self.handle = C.c_void_p()
        call("tbn_backbone_plan_create", cin, frames, h, w, C.byref(self.handle))
        self.frames = frames
        self.key = (frames, h, w)
        oh, ow, oc = C.c_int(), C.c_int(), C.c_int()
        call("tbn_backbone_out_shape", self.handle, C.byref(oh), C.byref(ow), C.byref(oc))
        self.out_shape = (frames, oh.value, ow.value, oc.value)
        self.pool = []  # [tensor, weakref-to-token or None]
        self.tuned = {True: False, False: False}   # per mode (training / eval)
        self.synced = {True: False, False: False}  # data parallel: this mode's choices are rank 0's (PlanSync)

    def workspace(self, training, device, token=None):
        need = lib().tbn_backbone_workspace_bytes(self.handle, int(training))
        for ent in self.pool:
            busy = ent[1] is not None and ent[1]() is not None
            if not busy and ent[0].numel() >= need and ent[0].device == device:
                ent[1] = weakref.ref(token) if token is not None else None
                return ent[0]
        buf = torch.empty(need, dtype=torch.uint8, device=device)
        self.pool.append([buf, weakref.ref(token) if token is not None else None])
        return buf

    def export_choices(self):
        """the tuned launch choices as bytes (include/tbn_hip.h: tbn_backbone_plan_export)"""
        n = lib().tbn_backbone_plan_export_bytes(self.handle)
        buf = C.create_string_buffer(n)
        call("tbn_backbone_plan_export", self.handle, buf, n)
        return buf.raw

    def import_choices(self, blob):
        call("tbn_backbone_plan_import", self.handle, C.c_char_p(bytes(blob)), len(blob))

    def clear_pairs(self):
        """drops every sibling-pair decision of the tuned plan (per GEMM 26 ints after the 8-int header: [4] eval forward,
        [12] training forward, [20] data gradient hold the pair flags): each 3x3 | double_3x3_1 pair then runs as its two
        tuned single launches.  Used where two programs must launch the SAME kernels to be compared bit for bit
        (tests/test_branch_gpu.py: the branch-mode program never pairs)"""
        import struct
        blob = bytearray(self.export_choices())
        for g in range((len(blob) - 32) // 104):
            for k in (4, 12, 20):
                struct.pack_into("<i", blob, 32 + 104 * g + 4 * k, 0)
        self.import_choices(bytes(blob))

    def fingerprint(self):
        return "%016x" % lib().tbn_backbone_plan_fingerprint(self.handle)

    # Opt-in plan cache (environment variable TBN_PLAN_CACHE=<directory>): the launch choices tuned by one process are
    # written there and adopted by later processes for the same problem and mode instead of tuning again (~10 s per
    # backbone and mode; and every run of a job then uses the SAME kernels).  Off by default: without the variable every
    # process tunes on its own box, which is what bench.py measures.  A file only counts for the library version that wrote
    # it (tbn_version) and is validated field by field on import (tbn_backbone_plan_import).
    def _cache_file(self, cin, training):
        d = os.environ.get("TBN_PLAN_CACHE")
        if not d:
            return None
        n, h, w = self.key
        return os.path.join(d, "tbnplan_v%d_c%d_f%d_%dx%d_%s.bin" % (lib().tbn_version(), cin, n, h, w,
                                                                    "train" if training else "eval"))

    def load_cached(self, cin, training):
        f = self._cache_file(cin, training)
        if f is None or not os.path.exists(f):
            return False
        cur = bytearray(self.export_choices())
        with open(f, "rb") as fh:
            blob = fh.read()
        if len(blob) != len(cur):
            return False
        # adopt only the fields this mode's tuning run decides (per GEMM 26 ints: [0:8] eval forward, [8:16] training
        # forward, [16:26] data gradient + weight-gradient tile): the other mode may have been tuned in this process already
        for g in range((len(cur) - 32) // 104):
            o = 32 + 104 * g
            lo, hi = (o + 32, o + 104) if training else (o, o + 32)
            cur[lo:hi] = blob[lo:hi]
        try:
            self.import_choices(bytes(cur))
        except TbnHipError:
            return False
        return True

    def store_cached(self, cin, training):
        f = self._cache_file(cin, training)
        if f is None:
            return
        os.makedirs(os.path.dirname(f), exist_ok=True)
        tmp = f + ".%d.tmp" % os.getpid()
        with open(tmp, "wb") as fh:
            fh.write(self.export_choices())
        os.replace(tmp, f)

    def __del__(self):
        try:
            if self.handle:
                lib().tbn_backbone_plan_destroy(self.handle)
        except Exception:
            pass


class _Token:
    """keeps a training workspace reserved until its autograd node is released"""
    pass


class _Flip:
    """state of the data-gradient weight copy of one training forward (BNInception.flip_weights_early)"""
    __slots__ = ("plan", "ws", "weight", "version", "done", "__weakref__")


class _BackboneFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight, bias, g0, g1, b0, b1, module, freq_only):
        plan = module._plan(x.shape[0], x.shape[2], x.shape[3])
        training = module.training
        # (grad mode is off inside Function.forward; needs_input_grad already reflects no_grad callers)
        need_grad = training and any(ctx.needs_input_grad[1:7])
        token = _Token() if need_grad else None
        ws = plan.workspace(training, x.device, token)
        gamma = torch.cat([g0, g1]) if g1.numel() else g0
        beta = torch.cat([b0, b1]) if b1.numel() else b0
        prm = BackboneParams(ptr(weight), ptr(bias), ptr(gamma), ptr(beta), ptr(module.running_mean),
                             ptr(module.running_var), 0.1, 1e-5, module._side_stream_ptr(), module._engine_flags())
        feat_ptr = C.c_void_p()
        st = stream_ptr()
        sync = module.plan_sync
        if module.autotune and (not plan.tuned[training] or (sync is not None and not plan.synced[training])):
            # first use of this (shape, mode): run once so every buffer holds real data, time the tile
            # candidates of each layer on it, then do the real forward below with the tuned plan.
            # Data parallel (`plan_sync` set by core.models.DataParallel): only rank 0 tunes; its choices travel to
            # every replica (tbn_backbone_plan_export / _import), so that all ranks run the SAME kernels -- per-rank
            # tuning from noisy on-box timings would let replicas differ and make the job's step the slowest plan's.
            # The exchange is a COLLECTIVE every rank must enter for the same (problem, mode): PlanSync.check gathers the
            # key first and raises on every rank when they differ (round-5 advisor); a rank that tuned this plan before it
            # was wrapped still takes part (`synced`, not `tuned`, decides) and adopts rank 0's choices.
            if sync is not None:
                sync.check((module.in_channels,) + plan.key + (int(training),), x.device)
            tune_here = not plan.tuned[training] and (sync is None or sync.is_source())
            cached = plan.load_cached(module.in_channels, training) if tune_here else False
            if cached:
                pass                     # TBN_PLAN_CACHE: choices of an earlier process (same library version, same problem)
            elif tune_here:
                saved = (module.running_mean.clone(), module.running_var.clone())
                call("tbn_backbone_forward", plan.handle, int(training), ptr(x), C.byref(prm), ptr(ws), ws.numel(),
                     C.byref(feat_ptr), st)
                call("tbn_backbone_autotune", plan.handle, int(training), C.byref(prm), ptr(ws), ws.numel(), st)
                module.running_mean.copy_(saved[0])
                module.running_var.copy_(saved[1])
                plan.store_cached(module.in_channels, training)
            if sync is not None:
                plan.import_choices(sync.broadcast(plan.export_choices(), x.device))
                plan.synced[training] = True
            plan.tuned[training] = True
        call("tbn_backbone_forward", plan.handle, int(training), ptr(x), C.byref(prm), ptr(ws), ws.numel(),
             C.byref(feat_ptr), st)
        N, H, W, Cc = plan.out_shape
        slot = module._out_slot
        module._out_slot = None
        if (not freq_only and slot is not None and tuple(slot.shape) == (N, Cc) and slot.stride(1) == 1 and slot.is_cuda
                and slot.device == x.device and slot.dtype == torch.float32):
            out = slot           # the caller's column range of a wider buffer (TBNModel: the concatenated features)
        else:
            out = torch.empty((N, W, Cc) if freq_only else (N, Cc), device=x.device, dtype=torch.float32)
        call("tbn_spatial_mean_fwd", feat_ptr.value, Cc, ptr(out), Cc if freq_only else out.stride(0), N, H, W, Cc,
             int(freq_only), st)
        if training:
            module.num_batches_tracked += 1
        if need_grad:
            ctx.plan, ctx.ws, ctx.token, ctx.module, ctx.freq_only = plan, ws, token, module, freq_only
            ctx.save_for_backward(weight, bias, gamma, beta)
            f = ctx.flip = _Flip()
            f.plan, f.ws, f.weight, f.version, f.done = plan, ws, weight, weight._version, False
            module._last_flip = weakref.ref(f)
        ctx.need_grad = need_grad
        ctx.eval_mode = not training
        return out

    @staticmethod
    def backward(ctx, dout):
        if ctx.needs_input_grad[0]:
            raise TbnHipError("BNInception: the gradient with respect to the input frames is not implemented (the "
                              "first conv's data gradient is never formed); detach the input")
        if not ctx.need_grad:
            if ctx.eval_mode and any(ctx.needs_input_grad[1:7]):
                raise TbnHipError("BNInception: backward through an eval-mode backbone (running-statistics BN) is not "
                                  "implemented; call .train() for fine-tuning or wrap inference in torch.no_grad()")
            return (None,) * 9
        weight, bias, gamma, beta = ctx.saved_tensors
        plan, ws, module = ctx.plan, ctx.ws, ctx.module
        N, H, W, Cc = plan.out_shape
        st = stream_ptr()
        if ctx.freq_only or dout.dim() != 2 or dout.stride(1) != 1 or dout.dtype != torch.float32:
            dout = dout.contiguous().float()
        dout_ld = Cc if ctx.freq_only else dout.stride(0)      # a column range of the fusion layer's input gradient: read in place
        dfeat = torch.empty(plan.out_shape, device=dout.device, dtype=torch.float32)
        call("tbn_spatial_mean_bwd", ptr(dout), dout_ld, ptr(dfeat), Cc, N, H, W, Cc, int(ctx.freq_only), st)
        dw = torch.empty_like(weight)
        db = torch.empty_like(bias)
        need = ctx.needs_input_grad
        bn_first, bn_rest = need[3] or need[5], need[4] or need[6]
        dg = torch.zeros_like(gamma) if (bn_first or bn_rest) else None
        dbe = torch.zeros_like(beta) if (bn_first or bn_rest) else None
        # the data-gradient weight copy was made early (flip_weights_early) and the weights are still the ones it read
        flipped = ctx.flip.done and ctx.flip.version == weight._version
        prm = BackboneParams(ptr(weight), ptr(bias), ptr(gamma), ptr(beta), ptr(module.running_mean),
                             ptr(module.running_var), 0.1, 1e-5, module._side_stream_ptr(),
                             module._engine_flags() | (4 if flipped else 0))
        aux = 0
        # inside a stream capture (torch.cuda.graph of a whole step) the weight gradients stay on the launch stream: the
        # engine refuses an aux stream there (TBN_ERR_UNSUPPORTED -- a capture that forks from an already forked stream
        # overflows the stack of ROCm 7.x's hipStreamEndCapture, and whether this stream is the capture's origin cannot be
        # queried), so a captured step degrades to serial weight-gradient launches instead of raising from autograd
        if module.use_aux_stream and not torch.cuda.is_current_stream_capturing():
            cur = torch.cuda.current_stream()
            a = module._aux_streams.get(cur.cuda_stream)
            if a is None:
                a = module._aux_streams[cur.cuda_stream] = torch.cuda.Stream(device=dout.device)
            aux = a.cuda_stream
        # Gradient buckets (data parallel; include/tbn_hip.h tbn_backbone_grads.bucket_cb): the engine calls back as soon as
        # the weight gradients of inception_5a..5b, then of 4a..4e, are enqueued; `grad_bucket_fn` (set by DataParallel)
        # starts the all-reduce of that slice of `dw` behind this stream, so it runs under the 3x / stem layers of the pass.
        # A handle it returns is waited for (a stream-level wait under RCCL) before `dw` is handed to autograd.
        bucket_fn = module.grad_bucket_fn if need[1] else None
        waits, failure = [], []

        def on_bucket(_user, first, count):
            try:                                   # an exception must not unwind through the C frames of the engine
                w = bucket_fn(module.flat_weight, dw, int(first), int(first) + int(count))
                if w is not None:
                    waits.append(w)
            except BaseException as e:             # noqa: B902 -- re-raised below, after the engine call has returned
                failure.append(e)

        cb = BUCKET_CB(on_bucket) if bucket_fn is not None else BUCKET_CB()
        grads = BackboneGrads(ptr(dw), ptr(db), ptr(dg), ptr(dbe), 2 if bn_rest else (1 if bn_first else 0), aux, cb, 0)
        call("tbn_backbone_backward", plan.handle, ptr(dfeat), C.byref(prm), C.byref(grads), ptr(ws), ws.numel(), st)
        ctx.token = None  # workspace may be reused now
        for w in waits:
            w()
        if failure:
            raise failure[0]
        n0 = module.first_bn_channels
        return (None, dw if need[1] else None, db if need[2] else None,
                dg[:n0] if need[3] else None, dg[n0:] if need[4] else None,
                dbe[:n0] if need[5] else None, dbe[n0:] if need[6] else None, None, None)


class BNInception(nn.Module):
    """Drop-in for the reference's `BNInception` (pretrainedmodels graph + logits override)."""

    def __init__(self, num_classes=1000, in_channels=3):
        super().__init__()
        self.in_channels = in_channels
        self.num_classes = num_classes
        self.is_audio = False
        self.attend = False
        self.feature_size = 1024
        self.eval_chunk = 256       # frames per engine call in eval mode (bounds the workspace)
        self.autotune = True        # time the GEMM tile candidates per layer on first use of a shape
        self.use_aux_stream = True  # weight-gradient GEMMs on a second HIP stream (overlaps dgrad / BN backward)
        self._aux_streams = {}
        # branch mode: the 3x3 / pool_proj chain of every inception block on a side stream beside the 1x1 -> double_3x3
        # chain (include/tbn_hip.h, tbn_backbone_params.side_stream); pays while this backbone is the only one running
        self.use_branch_streams = True
        self._side_streams = {}
        # riders: the BN apply / BN-backward apply of a block's independent column ranges run inside the grid of a sibling
        # GEMM launch (include/tbn_hip.h TBN_BACKBONE_RIDERS); one-chain program only -- the engine ignores the flag in
        # branch mode -- and bit-identical to the stand-alone passes
        self.use_riders = True
        self.stem_wgrad_last = False    # TBN_BACKBONE_STEM_WGRAD_LAST: see include/tbn_hip.h (bench.py --stem-wgrad-last)
        self._last_flip = None          # weak reference to the last training forward's _Flip (flip_weights_early)
        self._out_slot = None       # set by TBNModel for one forward: where the pooled (frames, 1024) feature is to be written
        self.plan_sync = None       # data parallel: object with is_source() / check(key, device) / broadcast(blob, device)
                                    # (DataParallel.PlanSync).  A forward that only SOME ranks run (validation on rank 0)
                                    # must set it to None first: the exchange is a collective
        self.grad_bucket_fn = None  # data parallel: fn(param, fresh_grad, lo, hi) -> waiter or None, called from inside backward
                                    # as slices of the flat weight gradient become final (DataParallel._on_grad_bucket)
        self._plans = OrderedDict()
        # layer table from the engine (needs the library, not a GPU)
        probe = C.c_void_p()
        call("tbn_backbone_plan_create", in_channels, 1, 64, 64, C.byref(probe))
        self._layers = OrderedDict()
        info = ConvInfo()
        for i in range(lib().tbn_backbone_num_convs(probe)):
            call("tbn_backbone_conv_info", probe, i, C.byref(info))
            self._layers[info.name.decode()] = dict(cin=info.cin, cout=info.cout, k=info.ksize, stride=info.stride,
                                                    pad=info.pad, w_off=info.weight_offset, c_off=info.channel_offset)
        nw, nc = lib().tbn_backbone_weight_floats(probe), lib().tbn_backbone_channel_floats(probe)
        lib().tbn_backbone_plan_destroy(probe)
        order = reference_conv_order()
        assert sorted(order) == sorted(self._layers), "engine layer table does not match the reference graph"
        self._order = order
        assert self._layers["conv1_7x7_s2"]["c_off"] == 0
        self.first_bn_channels = self._layers["conv1_7x7_s2"]["cout"]
        n0 = self.first_bn_channels
        self.flat_weight = nn.Parameter(torch.empty(nw))
        self.flat_bias = nn.Parameter(torch.empty(nc))
        self.bn_weight_first = nn.Parameter(torch.ones(n0))
        self.bn_weight_rest = nn.Parameter(torch.ones(nc - n0))
        self.bn_bias_first = nn.Parameter(torch.zeros(n0))
        self.bn_bias_rest = nn.Parameter(torch.zeros(nc - n0))
        self.register_buffer("running_mean", torch.zeros(nc))
        self.register_buffer("running_var", torch.ones(nc))
        self.register_buffer("num_batches_tracked", torch.zeros(len(order), dtype=torch.long))
        self.last_linear = nn.Linear(1024, num_classes)  # dropped by the factory, as in the reference
        self.reset_parameters()

    # ------------------------------------------------------------------ parameters
    def reset_parameters(self):
        with torch.no_grad():
            for name, L in self._layers.items():
                w = self.conv_weight(name)
                nn.init.kaiming_uniform_(w, a=5 ** 0.5)
                bound = 1.0 / (L["cin"] * L["k"] * L["k"]) ** 0.5
                self.flat_bias[L["c_off"]:L["c_off"] + L["cout"]].uniform_(-bound, bound)

    def conv_weight(self, name):
        """OIHW view (channels_last memory) of one conv's weights inside the flat buffer"""
        L = self._layers[name]
        n = L["cout"] * L["k"] * L["k"] * L["cin"]
        return self.flat_weight.data[L["w_off"]:L["w_off"] + n].view(L["cout"], L["k"], L["k"], L["cin"]).permute(
            0, 3, 1, 2)

    def _chan(self, name, flat_first, flat_rest=None):
        L = self._layers[name]
        a, b = L["c_off"], L["c_off"] + L["cout"]
        if flat_rest is None:
            return flat_first.data[a:b]
        n0 = self.first_bn_channels
        return flat_first.data[a:b] if b <= n0 else flat_rest.data[a - n0:b - n0]

    def reference_param_views(self, flat):
        """(key, view) pairs of the four PARAMETER kinds in the reference's order, taken from `flat` =
        {attribute name: tensor shaped like that flat parameter} -- e.g. the momentum buffers of the optimiser"""
        n0 = self.first_bn_channels
        for name in self._order:
            L = self._layers[name]
            n = L["cout"] * L["k"] * L["k"] * L["cin"]
            a, b = L["c_off"], L["c_off"] + L["cout"]
            yield name + ".weight", flat["flat_weight"][L["w_off"]:L["w_off"] + n].view(
                L["cout"], L["k"], L["k"], L["cin"]).permute(0, 3, 1, 2)
            yield name + ".bias", flat["flat_bias"][a:b]
            first = b <= n0
            yield name + "_bn.weight", (flat["bn_weight_first"][a:b] if first else flat["bn_weight_rest"][a - n0:b - n0])
            yield name + "_bn.bias", (flat["bn_bias_first"][a:b] if first else flat["bn_bias_rest"][a - n0:b - n0])

    def named_reference_tensors(self):
        """(key, tensor view) pairs in the reference's state_dict order"""
        for i, name in enumerate(self._order):
            yield name + ".weight", self.conv_weight(name)
            yield name + ".bias", self._chan(name, self.flat_bias)
            yield name + "_bn.weight", self._chan(name, self.bn_weight_first, self.bn_weight_rest)
            yield name + "_bn.bias", self._chan(name, self.bn_bias_first, self.bn_bias_rest)
            yield name + "_bn.running_mean", self._chan(name, self.running_mean)
            yield name + "_bn.running_var", self._chan(name, self.running_var)
            yield name + "_bn.num_batches_tracked", self.num_batches_tracked[i]

    def _save_to_state_dict(self, destination, prefix, keep_vars):
        for k, v in self.named_reference_tensors():
            destination[prefix + k] = v if keep_vars else v.detach().clone(memory_format=torch.contiguous_format)
        # nn.Module.state_dict() recurses into children (last_linear) by itself

    def _load_from_state_dict(self, state_dict, prefix, local_metadata, strict, missing_keys, unexpected_keys,
                              error_msgs):
        own = set()
        with torch.no_grad():
            for k, dst in self.named_reference_tensors():
                key = prefix + k
                own.add(key)
                if key not in state_dict:
                    if strict and not k.endswith("num_batches_tracked"):
                        missing_keys.append(key)
                    continue
                src = state_dict[key]
                if tuple(src.shape) != tuple(dst.shape):
                    error_msgs.append(f"size mismatch for {key}: copying a param with shape {tuple(src.shape)} "
                                      f"from checkpoint, the shape in current model is {tuple(dst.shape)}.")
                    continue
                dst.copy_(src)
        if strict:
            child_prefixes = tuple(prefix + n + "." for n, _ in self.named_children())
            for key in state_dict:
                if key.startswith(prefix) and key not in own and not key.startswith(child_prefixes):
                    unexpected_keys.append(key)

    def set_bn_trainable(self, first, rest):
        """BN affine requires_grad split used by TBNModel 'partialbn' (reference model.py:164-176)"""
        self.bn_weight_first.requires_grad = self.bn_bias_first.requires_grad = first
        self.bn_weight_rest.requires_grad = self.bn_bias_rest.requires_grad = rest

    # ------------------------------------------------------------------ compute
    def _plan(self, frames, h, w):
        key = (frames, h, w)
        if key not in self._plans:
            if len(self._plans) >= 4:
                self._plans.popitem(last=False)
            self._plans[key] = _Plan(self.in_channels, frames, h, w)
        else:
            self._plans.move_to_end(key)
        return self._plans[key]

    def plan_fingerprints(self):
        """{(frames, H, W): hex fingerprint of the plan's launch choices} -- equal across the replicas of a data-parallel
        job once `plan_sync` is set; `bench.py` prints them so that a slow box can be told from a different plan"""
        return {k: p.fingerprint() for k, p in self._plans.items()}

    def flip_weights_early(self):
        """Launch, on the CURRENT stream (which must be the one the last training forward of this module ran on), the
        flipped / transposed data-gradient weight copy its backward pass starts with (tbn_backbone_flip_weights,
        include/tbn_hip.h).  TBNModel calls this right after joining the modality streams for the heads: the copy
        (~0.1 ms per backbone, HBM-bound) then runs beside the heads' ~20 small dependent launches instead of between
        them and the first backward GEMM.  The backward pass skips its own copy only if the weights are unchanged
        (`_version`); no-op when the last forward needs no backward or while a graph is being captured."""
        f = self._last_flip() if getattr(self, "_last_flip", None) is not None else None
        self._last_flip = None
        if f is None or f.done or f.weight._version != f.version or torch.cuda.is_current_stream_capturing():
            return False
        prm = BackboneParams(ptr(f.weight), 0, 0, 0, 0, 0, 0.1, 1e-5, 0, 0)
        call("tbn_backbone_flip_weights", f.plan.handle, C.byref(prm), ptr(f.ws), f.ws.numel(), stream_ptr())
        f.done = True
        return True

    def _engine_flags(self):
        return (1 if self.use_riders else 0) | (2 if self.stem_wgrad_last else 0)   # TBN_BACKBONE_RIDERS | TBN_BACKBONE_STEM_WGRAD_LAST

    def _side_stream_ptr(self):
        """the side stream that goes with the current stream (0: serial program; also while a graph is being captured --
        the engine would ignore it there anyway)"""
        if not self.use_branch_streams or torch.cuda.is_current_stream_capturing():
            return 0
        cur = torch.cuda.current_stream()
        s = self._side_streams.get(cur.cuda_stream)
        if s is None:
            s = self._side_streams[cur.cuda_stream] = torch.cuda.Stream(device=cur.device)
        return s.cuda_stream

    def _run(self, x, freq_only):
        if not x.is_cuda:
            raise TbnHipError("BNInception: input is on the CPU; this backbone only runs on an MI355X via "
                              "libtbn_hip.so (no CPU fallback)")
        x = x.contiguous().float()
        args = (self.flat_weight, self.flat_bias, self.bn_weight_first, self.bn_weight_rest, self.bn_bias_first,
                self.bn_bias_rest, self, freq_only)
        if self.training or x.shape[0] <= self.eval_chunk:
            return _BackboneFn.apply(x, *args)
        outs = [_BackboneFn.apply(x[i:i + self.eval_chunk], *args) for i in range(0, x.shape[0], self.eval_chunk)]
        return torch.cat(outs, 0)

    def forward(self, x):
        """(frames, C, H, W) -> pooled features exactly as reference logits(features(x)):
        (frames, 1024), or (frames, 1024, 1, T) for attended audio"""
        freq = bool(self.is_audio and self.attend)
        out = self._run(x, freq)
        if freq:  # reference layout (N, 1024, 1, T); memory stays (N, T, 1024)
            return out.permute(0, 2, 1).unsqueeze(2)
        return out

    def forward_sequence(self, x):
        """attended-audio features in the engine's native (frames, T, 1024) layout"""
        return self._run(x, True)

    def features(self, x):
        """reference-shaped (frames, 1024, h, w) feature map (API parity; copies out of the workspace)"""
        x = x.contiguous().float()
        plan = self._plan(x.shape[0], x.shape[2], x.shape[3])
        ws = plan.workspace(False, x.device)
        gamma = torch.cat([self.bn_weight_first, self.bn_weight_rest])
        beta = torch.cat([self.bn_bias_first, self.bn_bias_rest])
        if self.training:
            raise TbnHipError("features(): only available in eval mode; use forward() for training")
        prm = BackboneParams(ptr(self.flat_weight), ptr(self.flat_bias), ptr(gamma), ptr(beta),
                             ptr(self.running_mean), ptr(self.running_var), 0.1, 1e-5)
        feat_ptr = C.c_void_p()
        call("tbn_backbone_forward", plan.handle, 0, ptr(x), C.byref(prm), ptr(ws), ws.numel(), C.byref(feat_ptr),
             stream_ptr())
        N, H, W, Cc = plan.out_shape
        off = (feat_ptr.value - ws.data_ptr()) // 4
        nhwc = ws.view(torch.float32)[off:off + N * H * W * Cc].view(N, H, W, Cc).clone()
        return nhwc.permute(0, 3, 1, 2)

    def logits(self, features):
        """reference bn_inception.py:16-35 on a (frames, 1024, h, w) tensor"""
        nhwc = features.permute(0, 2, 3, 1).contiguous()
        if self.is_audio and self.attend:
            return ops.spatial_mean(nhwc, True).permute(0, 2, 1).unsqueeze(2)
        return ops.spatial_mean(nhwc, False)


def bninception(in_channels, modality, pretrained="imagenet", model_dir="", is_audio=False, attend=False,
                state_dict=None):
    """reference factory core/models/bn_inception.py:38-107.

    `state_dict` lets a caller hand over the weights directly instead of the Drive-hosted
    `weights/{imagenet_bninception_rgb,kinetics_bninception_flow}.pth` files; with `pretrained`
    set and no `state_dict` the file is loaded from `model_dir` exactly like the reference.
    """
    num_classes = 1000
    data_dict = state_dict
    if pretrained is not None:
        if pretrained == "kinetics":
            num_classes = 400
            file = os.path.join(model_dir, "kinetics_bninception_flow.pth")
        elif pretrained == "imagenet":
            file = os.path.join(model_dir, "imagenet_bninception_rgb.pth")
        if data_dict is None:
            data_dict = torch.load(file, map_location="cpu")
    model = BNInception(num_classes=num_classes, in_channels=in_channels)
    if data_dict is not None:
        data_dict = dict(data_dict)
        if modality == "Audio":
            # first conv = channel mean of the RGB filter; reconcile any other key differences
            data_dict["conv1_7x7_s2.weight"] = data_dict["conv1_7x7_s2.weight"].mean(dim=1).unsqueeze(dim=1)
            own = model.state_dict()
            for k in [k for k in own.keys() if k not in data_dict]:
                data_dict[k] = own[k]
            for k in [k for k in data_dict.keys() if k not in own]:
                del data_dict[k]
        model.load_state_dict(data_dict)
        print(f"Model initialized with {pretrained} weights")
    model.is_audio = is_audio
    model.attend = attend
    model.feature_size = 1024
    delattr(model, "last_linear")
    return model
