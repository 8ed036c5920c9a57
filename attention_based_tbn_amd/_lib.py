"""ctypes binding of libtbn_hip.so (include/tbn_hip.h).  There is NO CPU fallback: if the
library is missing or a call fails, the product path raises."""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libtbn_hip.so")
if os.environ.get("TBN_LIB"):
    # diagnostic / A-B builds (attention_based_tbn_amd/build.py, TBN_BUILD_VARIANT): loaded INSTEAD of the shipped library,
    # never copied over it; said out loud because results of a TBN_DIAG build are invalid
    import sys as _sys
    LIB_PATH = os.path.abspath(os.environ["TBN_LIB"])
    print(f"[tbn] TBN_LIB set: loading {LIB_PATH} instead of the shipped libtbn_hip.so", file=_sys.stderr)

c_fp = C.c_void_p   # device pointers are passed as integers (tensor.data_ptr())
c_i = C.c_int
c_sz = C.c_size_t
c_f = C.c_float


class ConvInfo(C.Structure):
    _fields_ = [("name", C.c_char * 64), ("cin", c_i), ("cout", c_i), ("ksize", c_i), ("stride", c_i),
                ("pad", c_i), ("weight_offset", c_sz), ("channel_offset", c_sz)]


class BackboneParams(C.Structure):
    _fields_ = [("weight", c_fp), ("bias", c_fp), ("gamma", c_fp), ("beta", c_fp), ("running_mean", c_fp),
                ("running_var", c_fp), ("momentum", c_f), ("eps", c_f), ("side_stream", c_fp), ("flags", c_i)]


BUCKET_CB = C.CFUNCTYPE(None, C.c_void_p, C.c_size_t, C.c_size_t)   # tbn_backbone_grads.bucket_cb(user, first_float, num_floats)


class BackboneGrads(C.Structure):
    _fields_ = [("dweight", c_fp), ("dbias", c_fp), ("dgamma", c_fp), ("dbeta", c_fp), ("bn_grad_layers", c_i),
                ("aux_stream", c_fp), ("bucket_cb", BUCKET_CB), ("bucket_user", c_fp)]


class ConvRed(C.Structure):
    _fields_ = [("y", c_fp), ("y_ld", c_i), ("col_begin", c_i), ("channels", c_i), ("stat_offset", c_i),
                ("partial", c_fp)]


class ConvDesc(C.Structure):
    """include/tbn_hip.h tbn_conv_desc"""
    _fields_ = [("inp", c_fp), ("in_ld", c_i), ("weight", c_fp), ("bias", c_fp), ("out", c_fp), ("out_ld", c_i),
                ("n", c_i), ("h", c_i), ("w", c_i), ("cin", c_i), ("cout", c_i), ("ksize", c_i), ("stride", c_i),
                ("pad", c_i), ("dgrad", c_i), ("epilogue", c_i), ("flags", c_i), ("stages", c_i), ("scale", c_fp),
                ("shift", c_fp), ("stat_partial", c_fp), ("nred", c_i), ("red", ConvRed * 4), ("red_stats", c_fp),
                ("red_stats_stride", c_i)]


class OptTensor(C.Structure):
    _fields_ = [("param", c_fp), ("grad", c_fp), ("momentum", c_fp), ("count", c_sz)]


# name -> (restype, argtypes); every symbol declared in include/tbn_hip.h
SIGNATURES = {
    "tbn_version": (c_i, []),
    "tbn_last_error": (C.c_char_p, []),
    "tbn_profile_enable": (c_i, [c_i]),
    "tbn_profile_reset": (c_i, []),
    "tbn_profile_num_entries": (c_i, []),
    "tbn_profile_entry": (c_i, [c_i, C.c_char_p, c_i, C.POINTER(C.c_long), C.POINTER(C.c_double),
                                C.POINTER(C.c_double)]),
    "tbn_profile_entry_bytes": (c_i, [c_i, C.POINTER(C.c_double)]),
    "tbn_timeline_enable": (c_i, [c_i]),
    "tbn_timeline_dump": (c_i, [C.c_char_p]),
    "tbn_diag_mfma_burst": (c_i, [c_fp, c_i, c_i, C.POINTER(C.c_double), c_fp]),
    "tbn_backbone_plan_create": (c_i, [c_i, c_i, c_i, c_i, C.POINTER(C.c_void_p)]),
    "tbn_backbone_plan_destroy": (None, [C.c_void_p]),
    "tbn_backbone_num_convs": (c_i, [C.c_void_p]),
    "tbn_backbone_conv_info": (c_i, [C.c_void_p, c_i, C.POINTER(ConvInfo)]),
    "tbn_backbone_weight_floats": (c_sz, [C.c_void_p]),
    "tbn_backbone_channel_floats": (c_sz, [C.c_void_p]),
    "tbn_backbone_workspace_bytes": (c_sz, [C.c_void_p, c_i]),
    "tbn_backbone_out_shape": (c_i, [C.c_void_p, C.POINTER(c_i), C.POINTER(c_i), C.POINTER(c_i)]),
    "tbn_backbone_num_streams": (c_i, [C.c_void_p]),
    "tbn_backbone_rider_launches": (c_i, [C.c_void_p, C.POINTER(c_i), C.POINTER(c_i)]),
    "tbn_backbone_tensor_info": (c_i, [C.c_void_p, C.c_char_p, c_i, C.POINTER(C.c_long), C.POINTER(c_i),
                                       C.POINTER(c_i), C.POINTER(c_i)]),
    "tbn_backbone_launch_info": (c_i, [C.c_void_p, C.c_char_p, c_i, C.POINTER(c_i)]),
    "tbn_backbone_plan_export_bytes": (c_sz, [C.c_void_p]),
    "tbn_backbone_plan_export": (c_i, [C.c_void_p, C.c_void_p, c_sz]),
    "tbn_backbone_plan_import": (c_i, [C.c_void_p, C.c_void_p, c_sz]),
    "tbn_backbone_plan_fingerprint": (C.c_ulonglong, [C.c_void_p]),
    "tbn_backbone_forward": (c_i, [C.c_void_p, c_i, c_fp, C.POINTER(BackboneParams), c_fp, c_sz,
                                   C.POINTER(C.c_void_p), c_fp]),
    "tbn_backbone_autotune": (c_i, [C.c_void_p, c_i, C.POINTER(BackboneParams), c_fp, c_sz, c_fp]),
    "tbn_backbone_backward": (c_i, [C.c_void_p, c_fp, C.POINTER(BackboneParams), C.POINTER(BackboneGrads), c_fp,
                                    c_sz, c_fp]),
    "tbn_backbone_flip_weights": (c_i, [C.c_void_p, C.POINTER(BackboneParams), c_fp, c_sz, c_fp]),
    "tbn_conv2d_fwd": (c_i, [c_fp, c_i, c_fp, c_fp, c_fp, c_i] + [c_i] * 10 + [c_fp, c_fp, c_fp, c_fp]),
    "tbn_conv2d_stat_tiles": (c_i, [c_i] * 8),
    "tbn_conv2d_fwd_tile": (c_i, [c_fp, c_i, c_fp, c_fp, c_fp, c_i] + [c_i] * 10 + [c_fp, c_i, c_i, c_fp]),
    "tbn_conv2d_dgrad": (c_i, [c_fp, c_i, c_fp, c_fp, c_i] + [c_i] * 9 + [c_fp, c_fp]),
    "tbn_conv_partial_rows": (c_i, [C.POINTER(ConvDesc), c_i, c_i]),
    "tbn_conv_launch": (c_i, [C.POINTER(ConvDesc), c_i, c_i, c_fp, c_fp]),
    "tbn_conv_launch_pair": (c_i, [C.POINTER(ConvDesc), C.POINTER(ConvDesc), c_i, c_i, c_i, c_fp, c_fp, c_fp]),
    "tbn_conv2d_wgrad_workspace_floats": (c_sz, [c_i] * 8),
    "tbn_conv2d_wgrad": (c_i, [c_fp, c_i, c_fp, c_i, c_fp] + [c_i] * 8 + [c_fp, c_fp]),
    "tbn_linear_fwd": (c_i, [c_fp, c_i, c_fp, c_fp, c_fp, c_i, c_i, c_i, c_i, c_i, c_fp]),
    "tbn_linear_dgrad": (c_i, [c_fp, c_i, c_fp, c_fp, c_i, c_i, c_i, c_i, c_i, c_fp, c_fp]),
    "tbn_linear_wgrad_workspace_floats": (c_sz, [c_i, c_i, c_i]),
    "tbn_linear_wgrad": (c_i, [c_fp, c_i, c_fp, c_i, c_fp, c_fp, c_i, c_i, c_i, c_fp, c_fp]),
    "tbn_bn_workspace_floats": (c_sz, [c_i, c_i]),
    "tbn_bn_relu_train_fwd": (c_i, [c_fp, c_i, c_i, c_fp, c_fp, c_fp, c_fp, c_f, c_f, c_fp, c_fp, c_fp, c_fp, c_fp,
                                    c_i, c_fp, c_fp]),
    "tbn_bn_relu_train_bwd": (c_i, [c_fp, c_i, c_fp, c_i, c_i, c_fp, c_fp, c_fp, c_fp, c_fp, c_fp, c_fp, c_fp, c_fp]),
    "tbn_bn_relu_maxpool_train_fwd": (c_i, [c_fp, c_i, c_i, c_i, c_i, c_fp, c_fp, c_fp, c_fp, c_f, c_f, c_fp, c_fp, c_fp,
                                            c_fp, c_fp, c_i, c_fp, c_i, c_i, c_i, c_i, c_fp, c_fp]),
    "tbn_bn_relu_maxpool_train_bwd": (c_i, [c_fp, c_i, c_fp, c_fp, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_fp, c_fp,
                                            c_fp, c_fp, c_fp, c_fp, c_fp, c_fp, c_fp]),
    "tbn_maxpool3_fwd": (c_i, [c_fp, c_i, c_fp, c_i, c_fp] + [c_i] * 8 + [c_fp]),
    "tbn_maxpool3_bwd": (c_i, [c_fp, c_i, c_fp, c_fp, c_i] + [c_i] * 9 + [c_fp]),
    "tbn_avgpool3_fwd": (c_i, [c_fp, c_i, c_fp, c_i] + [c_i] * 5 + [c_fp]),
    "tbn_spatial_mean_fwd": (c_i, [c_fp, c_i, c_fp, c_i] + [c_i] * 5 + [c_fp]),
    "tbn_spatial_mean_bwd": (c_i, [c_fp, c_i, c_fp, c_i] + [c_i] * 5 + [c_fp]),
    "tbn_pe_concat_fwd": (c_i, [c_fp, c_i, c_fp, c_fp, c_i, c_i, c_i, c_i, c_i, c_fp]),
    "tbn_groupnorm_fwd": (c_i, [c_fp, c_fp, c_fp, c_fp, c_fp, c_fp, c_i, c_i, c_i, c_i, c_f, c_fp]),
    "tbn_groupnorm_bwd": (c_i, [c_fp] * 8 + [c_i] * 4 + [c_fp]),
    "tbn_colsum": (c_i, [c_fp, c_i, c_fp, c_i, c_i, c_fp]),
    "tbn_mha_q1_fwd": (c_i, [c_fp] * 6 + [c_i] * 4 + [c_f, c_fp]),
    "tbn_mha_q1_bwd": (c_i, [c_fp] * 8 + [c_i] * 4 + [c_f, c_fp]),
    "tbn_weighted_sum_fwd": (c_i, [c_fp, c_fp, c_fp, c_i, c_i, c_i, c_i, c_fp]),
    "tbn_weighted_sum_bwd": (c_i, [c_fp, c_i, c_fp, c_fp, c_i, c_i, c_i, c_fp]),
    "tbn_segment_mean_fwd": (c_i, [c_fp, c_fp, c_i, c_i, c_i, c_fp]),
    "tbn_segment_mean_bwd": (c_i, [c_fp, c_fp, c_i, c_i, c_i, c_fp]),
    "tbn_dropout_fwd": (c_i, [c_fp, c_fp, c_f, c_fp, c_fp, c_sz, c_fp]),
    "tbn_ce_heads_fwd": (c_i, [c_fp, c_i, c_i, c_i, C.POINTER(c_i), C.POINTER(c_i), C.POINTER(C.c_void_p), c_fp, c_fp, c_fp, c_fp]),
    "tbn_ce_heads_bwd": (c_i, [c_fp, c_i, c_i, c_i, C.POINTER(c_i), C.POINTER(c_i), c_fp, c_fp, c_fp]),
    "tbn_mul_mask": (c_i, [c_fp, c_fp, c_fp, c_sz, c_fp]),
    "tbn_relu_mask_bwd": (c_i, [c_fp, c_fp, c_fp, c_fp, c_sz, c_fp]),
    "tbn_stft_twiddle_floats": (c_sz, []),
    "tbn_stft_make_twiddle": (c_i, [c_fp]),
    "tbn_stft_logpower": (c_i, [c_fp, c_i, c_i, c_fp, c_fp, c_f, c_fp]),
    "tbn_opt_num_partials": (c_i, [C.POINTER(OptTensor), c_i]),
    "tbn_opt_sqnorm_partials": (c_i, [C.POINTER(OptTensor), c_i, c_fp, c_fp]),
    "tbn_opt_clip_coef": (c_i, [c_fp, c_i, c_f, c_fp, c_fp, c_fp]),
    "tbn_opt_scale_grads": (c_i, [C.POINTER(OptTensor), c_i, c_fp, c_fp]),
    "tbn_opt_sgd_step": (c_i, [C.POINTER(OptTensor), c_i, c_f, c_f, c_f, c_fp, c_fp]),
    "tbn_topk_correct": (c_i, [c_fp, c_i, c_fp, c_i, c_i, c_i, c_fp, c_fp, c_fp, c_fp]),
    "tbn_frames_to_tensor": (c_i, [c_fp] + [c_i] * 16 + [c_fp, c_fp, c_i, c_i, c_fp, c_fp]),
}

_lib = None


class TbnHipError(RuntimeError):
    pass


def lib():
    """The loaded library; raises (never falls back) when it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise TbnHipError(
                f"{LIB_PATH} not found: build it with `python -m attention_based_tbn_amd.build` "
                "(hipcc --offload-arch=gfx950).  There is no CPU fallback for the TBN hot path.")
        # torch first: its wheel bundles its own HIP runtime (libamdhip64, SONAME .so.7 like the system's); whichever copy
        # is loaded first serves the whole process.  Loading libtbn_hip.so before torch pulled in /opt/rocm's runtime, and
        # the first kernel launch then failed with "no ROCm-capable device is detected" (the two disagree on the device
        # set-up torch does) -- seen when __graft_entry__.build() and smoke() ran in one process.
        import torch  # noqa: F401
        handle = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(handle, name)   # AttributeError if a declared symbol is not exported
            fn.restype = res
            fn.argtypes = args
        _lib = handle
    return _lib


def check(rc, what=""):
    if rc != 0:
        msg = lib().tbn_last_error()
        raise TbnHipError(f"{what} failed (rc={rc}): {msg.decode() if msg else ''}")


TRACE = None    # diagnostics (bench.py --trace-streams): a list collects (name, host t0, host t1, event0, event1) per
                # backbone forward / backward call, events recorded on the stream the call launches on


def call(name, *args):
    """Invoke a status-returning entry point and raise on error."""
    if TRACE is not None and name in ("tbn_backbone_forward", "tbn_backbone_backward"):
        import time
        import torch
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        t0 = time.perf_counter()
        check(getattr(lib(), name)(*args), name)
        t1 = time.perf_counter()
        e1.record()
        TRACE.append((name, t0, t1, e0, e1))
        return
    check(getattr(lib(), name)(*args), name)


def stream_ptr():
    import torch
    return torch.cuda.current_stream().cuda_stream


def ptr(t):
    return 0 if t is None else t.data_ptr()
