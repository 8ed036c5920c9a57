#!/usr/bin/env python
"""Headline benchmark: clips/s of the 3-segment RGB+Flow+Audio TBN training step (fwd+bwd)
on N MI355X GPUs of one node (BASELINE.json config 4: async sampling, attention off, 1.279 s
audio, batch 32 clips per GPU -- global batch 256 at 8 GPUs, i.e. weak scaling).

  python bench.py --gpus 1 --steps K --warmup W
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
         --master-port P bench.py --gpus N --steps K --warmup W

A step = forward + loss + backward (+ RCCL gradient all-reduce for N>1) + clip_grad_norm(20) +
SGD(momentum) update, exactly the reference loop body (core/tools/train.py:69-94), on synthetic
inputs already resident in HBM.  Prints ONE JSON line on rank 0.
"""
import argparse
import ctypes as C
import gc
import json
import os
import sys
import time

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

FLOP_PER_CLIP_FWD_BWD = 120.63e9      # BASELINE.md section 3 (convs only, 3 modalities x 3 segments)
# BASELINE.json configs[1..4]: overrides, batch per GPU, segments, mode, conv GFLOP per clip
CONFIGS = {
    2: dict(name="BASELINE config 2: RGB-only, attention off, 3 segments, 224x224, train step",
            ov=["data.flow.enable=False", "data.audio.enable=False", "model.attention.enable=False"],
            batch=32, train=True, gflop=35.86),
    3: dict(name="BASELINE config 3: RGB+Audio sync, MHA fusion + entropy loss, 3 segments, 1.279 s audio, train step",
            ov=["data.flow.enable=False", "data.audio.audio_length=1.279", "model.attention.use_entropy=True"],
            batch=64, train=True, gflop=81.47),
    4: dict(name="BASELINE config 4: RGB+Flow+Audio async, attention off, 3 segments, 224x224 frames "
                 "+ 1.279 s (256x256) spectrogram, train step fwd+loss+bwd+clip+SGD",
            ov=["model.attention.enable=False", "data.audio.audio_length=1.279", "data.sampling=async"],
            batch=32, train=True, gflop=120.63),
    5: dict(name="BASELINE config 5: RGB+Flow+Audio sync, MHA fusion, 25 test segments, eval forward + consensus",
            ov=["data.audio.audio_length=1.279"], batch=64, train=False, gflop=344.47),
}
PEAK_FP32_MFMA_TFLOPS = 157.3         # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, dense


def synthetic_batch(B, n, device, seed, modality=("RGB", "Flow", "Audio"), audio_w=256):
    g = torch.Generator(device=device).manual_seed(seed)
    mean = torch.tensor([0.408, 0.459, 0.502], device=device).view(1, 1, 3, 1, 1)
    inp = {}
    if "RGB" in modality:
        inp["RGB"] = torch.rand(B, n, 3, 224, 224, device=device, generator=g) - mean
    if "Flow" in modality:
        inp["Flow"] = torch.rand(B, n, 10, 224, 224, device=device, generator=g) - 0.502
    if "Audio" in modality:
        inp["Audio"] = (torch.randn(B, n, 1, 256, audio_w, device=device, generator=g) * 3 - 6).clamp_(-13.8155, 8.0)
    tgt = {"class": {"verb": torch.randint(0, 125, (B,), device=device, generator=g),
                     "noun": torch.randint(0, 352, (B,), device=device, generator=g)}}
    return inp, tgt


def collect_profile():
    from attention_based_tbn_amd._lib import lib
    L = lib()
    out = []
    name = C.create_string_buffer(96)
    for i in range(L.tbn_profile_num_entries()):
        n, ms, fl, by = C.c_long(), C.c_double(), C.c_double(), C.c_double()
        L.tbn_profile_entry(i, name, 96, C.byref(n), C.byref(ms), C.byref(fl))
        L.tbn_profile_entry_bytes(i, C.byref(by))
        out.append({"kernel": name.value.decode(), "launches": n.value, "ms": ms.value, "flops": fl.value,
                    "alg_bytes": by.value})
    return out


WORKLOAD_KEY = None    # set by main(): "config<N>_B<clips per GPU>[_fwd]" -- PMC traffic only counts for the workload it was collected on


def pmc_traffic(kernel):
    """HBM bytes per launch of `kernel` from the newest committed PMC summary (separate rocprofv3 --pmc
    FETCH_SIZE / WRITE_SIZE passes of this same command with the priming / autotune / warm-up steps filtered out,
    scripts/pmc_traffic.py); None if there is none.  PMC counters cannot be read from inside the process: the figure
    is the one collected from the commit named in the file's `note`."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_traffic.json")))
    if not files:
        return None
    try:
        with open(files[-1]) as f:
            doc = json.load(f)
        # a figure only counts for the kernels it was measured on: the file carries a hash of the kernel sources of its
        # tree (scripts/pmc_traffic.py); any change to them since makes it stale -> reported as such, never silently
        sys.path.insert(0, os.path.join(ROOT, "scripts"))
        from pmc_traffic import source_hash
        src = os.path.relpath(files[-1], ROOT)
        if doc.get("source_sha16") != source_hash(ROOT):
            return {"stale": True, "source": src,
                    "why": "kernel sources changed since these counters were collected (source_sha16 mismatch)"}
        # ... and for the WORKLOAD they were collected on: the per-name averages of a config-4 step (three modalities, B = 32)
        # describe a different launch mix than e.g. config 2's (round-4 advisor: a traffic_ratio of 0.988 came from that)
        measured_on = doc.get("workload", "config4_B32")
        if WORKLOAD_KEY is not None and measured_on != WORKLOAD_KEY:
            return {"stale": True, "source": src,
                    "why": f"counters were collected on workload {measured_on}, this run is {WORKLOAD_KEY}: the per-kernel "
                           "averages describe a different launch mix"}
        k = doc["kernels"].get(kernel)
        if k is None:
            return None
        out = {"hbm_bytes_per_launch": k["hbm_bytes_per_launch"], "read": k["hbm_read_bytes_per_launch"],
               "write": k["hbm_write_bytes_per_launch"], "source": src}
        if "calibration" in doc:
            # reads = L2 -> fabric requests x the bytes per request CALIBRATED on launches of known byte counts in this
            # kernel's access pattern (scripts/pmc_calibrate.py): the box's FETCH_SIZE formula prices a request at 64 B,
            # every calibration launch (wide streaming, BN passes, the weight gradient's 64-B row segments) measures 128
            out["bytes_per_read_request"] = k.get("bytes_per_request_used")
            out["read_requests_per_launch"] = k.get("rdreq_per_launch")
            out["calibration_bytes_per_request"] = {t: v["bytes_per_request_of_the_64B_class"] for t, v in doc["calibration"].items()}
        else:
            out["read_uncorrected"] = round(1024.0 * k["fetch_kib_raw_per_launch"])
        return out
    except (OSError, ValueError, KeyError, ImportError):
        return None


def _median(v):
    v = sorted(v)
    return v[len(v) // 2] if len(v) % 2 else 0.5 * (v[len(v) // 2 - 1] + v[len(v) // 2])


def roofline_object(samples, steps_in_first_sample, end_to_end_frac, sched=None):
    """`roofline` of the JSON line from the per-step profiler collections (instrumented single-stream steps run after the
    timed loop): the dominant kernel = the conv-GEMM instantiation with the largest summed time; `achieved` / `frac` and
    the conv-stage figure `all_conv_gemm` are MEDIANS over the samples, min / max beside them."""
    samples = [sm for sm in samples if sm]
    if not samples:
        return None
    conv = [[e for e in sm if not e["kernel"].startswith("linear: ")] for sm in samples]
    heads = [[e for e in sm if e["kernel"].startswith("linear: ")] for sm in samples]
    tot = {}
    for sm in conv:
        for e in sm:
            t = tot.setdefault(e["kernel"], {"kernel": e["kernel"], "launches": 0, "ms": 0.0, "flops": 0.0, "alg_bytes": 0.0})
            for k in ("launches", "ms", "flops", "alg_bytes"):
                t[k] += e[k]
    ranked = sorted(tot.values(), key=lambda e: -e["ms"])
    top = ranked[0]
    tf = lambda e: e["flops"] / (e["ms"] * 1e-3) / 1e12
    top_tf = [tf(e) for sm in conv for e in sm if e["kernel"] == top["kernel"]]
    top_us = [1e3 * e["ms"] / e["launches"] for sm in conv for e in sm if e["kernel"] == top["kernel"]]
    all_tf = [sum(e["flops"] for e in sm) / (sum(e["ms"] for e in sm) * 1e-3) / 1e12 for sm in conv]
    all_ms = [sum(e["ms"] for e in sm) for sm in conv]
    all_ms[0] /= steps_in_first_sample
    ach = _median(top_tf)
    tr = pmc_traffic(top["kernel"])
    tr_detail = tr
    if tr and tr.get("stale"):
        tr = None                      # stale counters: traffic is null, the detail says why
    alg_b = top["alg_bytes"] / max(1, top["launches"])
    fam = {}
    for e in conv[-1]:                 # launches per kernel family in ONE profiled step: which variants the per-layer
        f_ = e["kernel"].split("<")[0]  # autotuner actually put on this shape
        fam[f_] = fam.get(f_, 0) + e["launches"] // (steps_in_first_sample if len(conv) == 1 else 1)
    P = PEAK_FP32_MFMA_TFLOPS
    nstep_total = steps_in_first_sample if len(conv) == 1 else len(conv)     # profiled steps behind the totals
    # What the line LEADS with (round-5 verdict item 4): the conv stage as the TIMED schedule runs it -- conv FLOPs / time with
    # at least one conv GEMM running, all modality streams, from the library's own timeline -- not one kernel in isolation.
    # The dominant kernel's own figure (algorithmic FLOPs per launch / its average launch duration, HIP events on its
    # dispatch packet; the rocprofv3 summary under profiles/ agrees) sits under `dominant`, the one-stream conv stage under
    # `all_conv_gemm`.  Without timeline steps (--timeline-steps 0) the line leads with the one-stream conv stage.
    lead = sched if sched is not None else {"achieved": round(_median(all_tf), 2), "frac": round(_median(all_tf) / P, 4)}
    return {"bound": "mfma", "achieved": lead["achieved"], "peak": P, "unit": "TFLOP/s", "frac": lead["frac"],
            "frac_is": "conv_stage_timed_schedule" if sched is not None else "all_conv_gemm (one stream at a time)",
            "conv_stage_timed_schedule": sched,
            "dominant": {"kernel": top["kernel"], "achieved": round(ach, 2), "frac": round(ach / P, 4),
                         "frac_min": round(min(top_tf) / P, 4), "frac_max": round(max(top_tf) / P, 4),
                         "avg_launch_us": round(_median(top_us), 2), "launches": top["launches"] // nstep_total,
                         "alg_gflop_per_launch": round(top["flops"] / top["launches"] / 1e9, 4)},
            "samples": len(conv),
            "sampled": "instrumented single-stream steps AFTER the timed loop, one sample per step"
                       if steps_in_first_sample == 1 or len(conv) > 1 else
                       f"{steps_in_first_sample} instrumented steps INSIDE the timed loop (--profile-every), aggregated",
            "traffic": (tr or {}).get("hbm_bytes_per_launch"),
            "alg_bytes_per_launch": round(alg_b),
            "traffic_ratio": round(tr["hbm_bytes_per_launch"] / alg_b, 3) if tr and alg_b > 0 else None,
            "traffic_detail": tr_detail,
            "kernel_families": fam,
            "kernel": top["kernel"], "launches": top["launches"] // nstep_total,
            "avg_launch_us": round(_median(top_us), 2),
            "alg_gflop_per_launch": round(top["flops"] / top["launches"] / 1e9, 4),
            "all_conv_gemm": {"achieved": round(_median(all_tf), 2), "frac": round(_median(all_tf) / P, 4),
                              "frac_min": round(min(all_tf) / P, 4), "frac_max": round(max(all_tf) / P, 4),
                              "ms_per_profiled_step": round(_median(all_ms), 2)},
            "head_linear_gemm": {"launches": sum(e["launches"] for e in heads[-1]) // (steps_in_first_sample if len(conv) == 1 else 1),
                                 "ms_per_profiled_step": round(sum(e["ms"] for e in heads[-1]) /
                                                               (steps_in_first_sample if len(conv) == 1 else 1), 3)},
            "end_to_end_frac": round(end_to_end_frac, 4),
            "by_kernel": [{"kernel": e["kernel"], "launches": e["launches"] // nstep_total, "avg_us": round(1e3 * e["ms"] / e["launches"], 2),
                           "tflops": round(tf(e), 2)} for e in ranked[:8]]}


def host_cpu():
    """CPU model string, physical cores and hardware threads of this box (from /proc/cpuinfo)"""
    model, cores, threads = None, set(), 0
    try:
        phys = core = None
        with open("/proc/cpuinfo") as f:
            for ln in f:
                k, _, v = ln.partition(":")
                k, v = k.strip(), v.strip()
                if k == "model name" and model is None:
                    model = v
                elif k == "processor":
                    threads += 1
                elif k == "physical id":
                    phys = v
                elif k == "core id":
                    core = v
                elif not k and phys is not None and core is not None:
                    cores.add((phys, core))
                    phys = core = None
        if phys is not None and core is not None:
            cores.add((phys, core))
    except OSError:
        pass
    return {"cpu_model": model, "physical_cores": len(cores) or None, "hardware_threads": threads or os.cpu_count(),
            "usable_threads": len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else os.cpu_count()}


def query_rocm_smi():
    """power cap / clocks as rocm-smi reports them, or None.  Called FIRST THING in main(), before anything initialises the
    GPU in this process (rocm-smi is a child process: a fork + exec), and not at all under a profiler whose preloaded
    library has initialised the GPU before Python started (rocprofv3 --pmc): the GPU pool refuses an exec from there."""
    import subprocess
    if any(k in os.environ.get("LD_PRELOAD", "") for k in ("rocprof", "roctracer", "rocprofiler")) or any(
            k.startswith(("ROCPROF", "ROCP_", "ROCPROFILER")) for k in os.environ):
        return None
    try:
        r = subprocess.run(["rocm-smi", "--showpower", "--showmaxpower", "--showclocks", "--showperflevel", "--json"],
                           capture_output=True, text=True, timeout=15)
        doc = json.loads(r.stdout) if r.returncode == 0 and r.stdout.strip().startswith("{") else None
        if not doc:
            return None
        card = doc.get("card%d" % int(os.environ.get("LOCAL_RANK", "0"))) or next(iter(doc.values()))
        return {k: v for k, v in card.items()
                if any(t in k.lower() for t in ("power", "sclk", "mclk", "performance level"))}
    except (OSError, ValueError, subprocess.SubprocessError, StopIteration):
        return None


def box_identity(device, smi=None):
    """What this line was measured ON, so that two lines from two boxes of a pool explain their own gap: CPU model, GPU
    name, rocm-smi's power cap / clocks when the tool answers, and a warm pure-fp32-MFMA burst (tbn_diag_mfma_burst:
    registers only, no memory traffic) timed after 30 ms of the same load -- a cold burst runs ~12 % slow (profiles/HISTORY.md finding
    13) -- as TFLOP/s and as a fraction of the 157.3 the roofline is priced in."""
    from attention_based_tbn_amd._lib import call, ptr, stream_ptr
    out = dict(host_cpu())
    out["gpu_name"] = torch.cuda.get_device_name(device)
    props = torch.cuda.get_device_properties(device)
    out["gpu_cus"] = props.multi_processor_count
    out["gpu_arch"] = getattr(props, "gcnArchName", None)
    out["rocm_smi"] = smi
    sink = torch.zeros(16, device=device)
    fl = C.c_double()
    wg, iters = 1024, 1500                                     # ~2.6 ms per launch at the nominal peak
    for _ in range(14):                                        # ~35 ms of load: the clock is where it stays under a step
        call("tbn_diag_mfma_burst", ptr(sink), wg, iters, C.byref(fl), stream_ptr())
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    reps = 8
    for _ in range(reps):
        call("tbn_diag_mfma_burst", ptr(sink), wg, iters, C.byref(fl), stream_ptr())
    e1.record()
    torch.cuda.synchronize()
    tf = reps * fl.value / (e0.elapsed_time(e1) * 1e-3) / 1e12
    out["mfma_calibration"] = {"tflops": round(tf, 1), "frac_of_peak": round(tf / PEAK_FP32_MFMA_TFLOPS, 4),
                               "what": "v_mfma_f32_32x32x2_f32 register loop, 1024 workgroups x 4 waves, warm (after 35 ms "
                                       "of the same load), 8 launches of ~2.6 ms"}
    # Round 6: boxes of the pool run the bare MFMA loop within 0.5 % of each other and the training step up to 7 % apart
    # (profiles/r06_fast_box vs r06_slow_box) -- the step mixes MFMA work with ~60 GB of memory traffic.  Two more probes so that
    # a line can say which kind of box it ran on: a streaming copy (1 GiB read + 1 GiB written per launch, beyond the
    # Infinity Cache), alone, and the same MFMA burst WHILE that copy runs on a second stream.
    try:
        a = torch.empty(1 << 28, device=device, dtype=torch.float32)
        b = torch.empty_like(a)
        for _ in range(3):
            b.copy_(a)
        c0, c1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        c0.record()
        for _ in range(6):
            b.copy_(a)
        c1.record()
        torch.cuda.synchronize()
        copy_gbps = 6 * 2 * a.numel() * 4 / (c0.elapsed_time(c1) * 1e-3) / 1e9
        side = torch.cuda.Stream(device=device)
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(14):                                      # ~40 ms of copies beside the bursts below
                b.copy_(a)
        for _ in range(4):
            call("tbn_diag_mfma_burst", ptr(sink), wg, iters, C.byref(fl), stream_ptr())
        e0.record()
        for _ in range(reps):
            call("tbn_diag_mfma_burst", ptr(sink), wg, iters, C.byref(fl), stream_ptr())
        e1.record()
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        tf_mixed = reps * fl.value / (e0.elapsed_time(e1) * 1e-3) / 1e12
        out["memory_calibration"] = {"copy_GBps": round(copy_gbps, 0), "mfma_tflops_beside_the_copy": round(tf_mixed, 1),
                                     "what": "torch device copy of 1 GiB (read + written bytes counted), alone; the MFMA burst of "
                                             "mfma_calibration while such copies run on a second stream"}
        del a, b
    except RuntimeError as e:       # out of memory on a shared card (rehearsals with several ranks per GPU): not fatal
        out["memory_calibration"] = {"error": str(e)[:120]}
    return out


def _time_cpu(fn, warmup, min_steps, budget_s):
    """`warmup` untimed calls, then at least `min_steps` timed ones (more while the budget lasts, at most 3x)"""
    for _ in range(warmup):
        fn()
    t0 = time.perf_counter()
    n = 0
    while n < min_steps or (time.perf_counter() - t0 < budget_s and n < 3 * min_steps):
        fn()
        n += 1
    return n, time.perf_counter() - t0


def cpu_baseline():
    """the CPU oracle (torch-CPU fp32 restatement of the reference path, oneDNN) on this box's host
    cores, torch threads = physical cores, the same kind of seeded synthetic tensors as the GPU run, 2 warm-up + >= 5 timed
    iterations per leg:
      value        config-4 graph (RGB+Flow+Audio, attention off) at B = 4 clips x 3 segments, fwd + loss + bwd
      other[...]   the same graph forward only; config 1 exactly (Audio-only, (4,1,1,256,256)) fwd and fwd+bwd; the config-5
                   graph (RGB+Flow+Audio sync, MHA) at B = 1 clip x 25 segments, eval forward
    The oracle is the checker of this repo, timed here as the baseline -- never part of the product path."""
    from attention_based_tbn_amd.config import load_config, get_modality
    from oracle.fill import pretrained_pair
    from oracle.tbn import build_model as build_oracle
    cpu = host_cpu()
    old_threads = torch.get_num_threads()
    cores = min(cpu["physical_cores"] or old_threads, cpu["usable_threads"] or old_threads)
    torch.set_num_threads(max(1, cores))

    def make(ov, B, n, audio_w=256):
        cfg = load_config(ov)
        modality = get_modality(cfg)
        torch.manual_seed(0)
        model, crit, _ = build_oracle(cfg, modality, pretrained_pair(7))
        g = torch.Generator().manual_seed(0)
        inp = {}
        if "RGB" in modality:
            inp["RGB"] = torch.rand(B, n, 3, 224, 224, generator=g) - 0.45
        if "Flow" in modality:
            inp["Flow"] = torch.rand(B, n, 10, 224, 224, generator=g) - 0.5
        if "Audio" in modality:
            inp["Audio"] = (torch.randn(B, n, 1, 256, audio_w, generator=g) * 3 - 6).clamp_(-13.8155, 8.0)
        tgt = {"class": {"verb": torch.randint(0, 125, (B,), generator=g), "noun": torch.randint(0, 352, (B,), generator=g)}}
        return model, crit, inp, tgt

    def train_step(model, crit, inp, tgt):
        def f():
            model.zero_grad()
            out = model(inp)
            loss, _ = model.get_loss(crit, tgt, out, 0)
            loss["total"].backward()
        return f

    def fwd_only(model, inp):
        def f():
            with torch.no_grad():
                model(inp)
        return f

    legs = {}
    try:
        ov4 = ["model.attention.enable=False", "data.audio.audio_length=1.279", "data.sampling=async"]
        model, crit, inp, tgt = make(ov4, 4, 3)
        model.train()
        n, dt = _time_cpu(train_step(model, crit, inp, tgt), 2, 5, 20.0)
        head = {"value": 4 * n / dt, "steps": n, "threads": cores}
        sweep = {str(cores): round(head["value"], 4)}
        n, dt = _time_cpu(fwd_only(model, inp), 2, 5, 6.0)
        legs["config4_graph_B4_n3_train_mode_forward_only"] = {"clips_per_s": round(4 * n / dt, 3), "steps": n}
        # many-core hosts: one thread per physical core is what BASELINE.md section 4 prescribes, but oneDNN on a two-socket
        # box is several times FASTER with fewer threads (this graph at B = 4 is small).  `value` is the BEST thread count
        # measured (a baseline must not be a strawman, round-4 verdict); the prescribed leg stays beside it
        for th in (64, 32, 16):
            if th < cores:
                torch.set_num_threads(th)
                n, dt = _time_cpu(train_step(model, crit, inp, tgt), 1, 3, 5.0)
                sweep[str(th)] = round(4 * n / dt, 4)
                if 4 * n / dt > head["value"]:
                    head = {"value": 4 * n / dt, "steps": n, "threads": th}
        torch.set_num_threads(max(1, cores))
        legs["config4_graph_B4_n3_fwd_bwd_by_threads"] = sweep
        legs["config4_graph_B4_n3_fwd_bwd_prescribed_threads"] = {"clips_per_s": sweep[str(cores)], "threads": cores}
        del model
        ov1 = ["data.rgb.enable=False", "data.flow.enable=False", "model.attention.enable=False",
               "data.audio.audio_length=1.279", "train.num_segments=1"]
        model, crit, inp, tgt = make(ov1, 4, 1)
        model.train()
        n, dt = _time_cpu(train_step(model, crit, inp, tgt), 2, 5, 4.0)
        legs["config1_audio_only_B4_n1_fwd_bwd"] = {"clips_per_s": round(4 * n / dt, 3), "steps": n}
        n, dt = _time_cpu(fwd_only(model, inp), 2, 5, 3.0)
        legs["config1_audio_only_B4_n1_fwd"] = {"clips_per_s": round(4 * n / dt, 3), "steps": n}
        del model
        model, crit, inp, tgt = make(["data.audio.audio_length=1.279"], 1, 25)
        model.eval()
        n, dt = _time_cpu(fwd_only(model, inp), 2, 5, 10.0)
        legs["config5_graph_B1_n25_eval_fwd"] = {"clips_per_s": round(1 * n / dt, 3), "steps": n}
        del model
    finally:
        torch.set_num_threads(old_threads)
    return {"value": round(head["value"], 4), "unit": "clips/s", "cores": head["threads"], "kind": "port",
            "cpu_model": cpu["cpu_model"], "physical_cores": cpu["physical_cores"], "hardware_threads": cpu["hardware_threads"],
            "sample": f"{head['steps']} timed fwd+loss+bwd steps of the config-4 graph at B=4 clips x 3 segments, full-size "
                      f"seeded synthetic inputs, oracle/ (torch-CPU fp32, oneDNN) on the GPU box's host; best of the thread "
                      f"counts {sorted(int(k) for k in sweep)} = {head['threads']} threads (BASELINE.md section 4 prescribes one "
                      f"per physical core = {cores}: {sweep[str(cores)]} clips/s, under `other`)",
            "other": legs}


def self_launch(n):
    """`python bench.py --gpus N` without a launcher's environment: start the N ranks as a CHILD process
    (`python -m torch.distributed.run --nproc-per-node N bench.py <same arguments>`, rendezvous on 127.0.0.1), relay its
    stdout -- exactly one JSON line, from rank 0 -- and return its exit code.  This parent never touches the GPU (no
    torch.cuda call, no rocm-smi): the pool refuses an exec from a process that has, and a child is what it asks for."""
    import socket
    import subprocess
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")      # dmabuf IPC: RCCL / tensor sharing across the rank processes
    env.setdefault("OMP_NUM_THREADS", "4")
    print("[bench] no RANK / WORLD_SIZE in the environment: launching %d ranks: %s" % (n, " ".join(cmd)), file=sys.stderr)
    r = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, text=True)
    sys.stdout.write(r.stdout)
    sys.stdout.flush()
    return r.returncode


def timed_schedule_conv_stage(step, nsteps, conv_flops_per_step):
    """the conv stage AS THE TIMED SCHEDULE RUNS IT: `nsteps` more product steps (all streams, nothing serialised) under the
    library's own kernel timeline (tbn_timeline_enable: begin / end of every launch on its dispatch packet, ~2 % on the
    step, the modality streams keep overlapping -- an external tracer serialises them, profiles/HISTORY.md finding 17);
    achieved = conv FLOPs of those steps / the time during which at least one conv GEMM was running."""
    import csv
    import tempfile
    from attention_based_tbn_amd._lib import lib
    L = lib()
    torch.cuda.synchronize()
    L.tbn_timeline_enable(1)
    for _ in range(nsteps):
        step()
    torch.cuda.synchronize()
    L.tbn_timeline_enable(0)
    fd, path = tempfile.mkstemp(suffix=".csv", prefix="tbn_timeline_")
    os.close(fd)
    try:
        if L.tbn_timeline_dump(path.encode()) != 0:
            return None
        iv, first, last, n_all = [], None, None, 0
        with open(path) as f:
            for r in csv.DictReader(f):
                a, b = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
                n_all += 1
                first = a if first is None else min(first, a)
                last = b if last is None else max(last, b)
                nm = r["Kernel_Name"]
                if any(k in nm for k in ("conv_wgrad", "conv_igemm", "conv_halo", "conv_dma", "conv_pair", "conv_sk4")):
                    iv.append((a, b))
    finally:
        os.unlink(path)
    if not iv:
        return None
    iv.sort()
    union, summed, cur_a, cur_b = 0, 0, iv[0][0], iv[0][1]
    for a, b in iv:
        summed += b - a
        if a > cur_b:
            union += cur_b - cur_a
            cur_a, cur_b = a, b
        else:
            cur_b = max(cur_b, b)
    union += cur_b - cur_a
    tf = nsteps * conv_flops_per_step / (union * 1e-9) / 1e12
    return {"achieved": round(tf, 2), "frac": round(tf / PEAK_FP32_MFMA_TFLOPS, 4), "steps": nsteps,
            "gemm_running_ms_per_step": round(union / nsteps / 1e6, 3),
            "gemm_kernel_ms_per_step": round(summed / nsteps / 1e6, 3),
            "gemms_in_flight_while_running": round(summed / union, 2),
            "wall_ms_per_step": round((last - first) / nsteps / 1e6, 3), "launches_per_step": n_all // nsteps,
            "conv_gflop_per_step": round(conv_flops_per_step / 1e9, 1),
            "what": "conv FLOPs / time with >= 1 conv GEMM running, product steps (all modality streams) after the timed loop "
                    "under the library's kernel timeline (tbn_timeline_enable; costs the step ~2 %)"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--config", type=int, default=4, choices=sorted(CONFIGS),
                    help="BASELINE.json config (4 = the headline metric; 2, 3, 5 are the other single-GPU-sized configs)")
    ap.add_argument("--batch-per-gpu", type=int, default=0, help="clips per GPU (default: the config's)")
    ap.add_argument("--audio-2p1s", action="store_true",
                    help="config 3 only: the reference README's default 2.1 s audio window (256x420 spectrogram, T = 13) "
                         "instead of the metric's 1.279 s (256x256, T = 8)")
    ap.add_argument("--forward-only", action="store_true", help="time the forward pass only (training-mode BN)")
    ap.add_argument("--no-multi-stream", action="store_true", help="diagnostic: run the modality backbones on one stream")
    ap.add_argument("--no-aux-stream", action="store_true", help="diagnostic: weight gradients on the backbone's own stream")
    ap.add_argument("--aux-streams", default=None,
                    help="diagnostic: comma-separated modalities whose weight gradients run on a second stream "
                         "(default: the model's policy -- only a lone backbone)")
    ap.add_argument("--branch-streams", default=None,
                    help="diagnostic: comma-separated modalities (or 'all' / 'none') whose inception branches run on a side "
                         "stream (default: the model's policy -- only a lone backbone)")
    ap.add_argument("--riders", default=None, choices=["on", "off"],
                    help="diagnostic: BN apply / BN-backward apply passes riding in sibling GEMM launches (default: the model's)")
    ap.add_argument("--eval-chunk", type=int, default=0,
                    help="config 5: frames per engine call of the eval forward (default: the backbone's, 256)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--host-inputs", action="store_true",
                    help="diagnostic (never the headline): the batch starts in pinned HOST memory every step and is copied "
                         "over PCIe on a side stream, double-buffered against the previous step -- the PCIe-inclusive rate")
    ap.add_argument("--stft-inputs", action="store_true",
                    help="the audio leg starts from WAVEFORMS (B, n, 30695 samples = 1.279 s at 24 kHz, resident in HBM): the "
                         "log-power STFT kernel (reference dataset.py:483-495, CPU librosa there) runs inside the timed step")
    ap.add_argument("--profile-steps", type=int, default=3,
                    help="instrumented single-stream steps run AFTER the timed loop (kernel timestamps on every conv-GEMM "
                         "launch): `roofline` reports the median / min / max over them.  The timed region is the product "
                         "step only")
    ap.add_argument("--profile-every", type=int, default=0,
                    help="diagnostic (rocprofv3 / PMC passes): ALSO instrument every k-th step INSIDE the timed loop, so a "
                         "short traced run contains instrumented steps; the line then says timed_region_instrumented")
    ap.add_argument("--timeline", default=None, metavar="CSV",
                    help="diagnostic: after the timed loop run 4 more product steps with the library's kernel timeline on "
                         "(tbn_timeline_enable: begin / end of every launch on a common clock, no external tracer) and write "
                         "it to CSV for scripts/step_timeline.py -- the real overlap of the multi-stream step")
    ap.add_argument("--trace-streams", action="store_true",
                    help="diagnostic: after the timed loop run 3 more steps with HIP events around every backbone "
                         "forward / backward call and print (stderr) when each ran on the GPU and how long the host took "
                         "to issue it -- the stagger between the modality streams")
    ap.add_argument("--timeline-steps", type=int, default=4,
                    help="product steps run AFTER the timed loop under the library's kernel timeline: `roofline.frac` is the "
                         "conv stage of that schedule (0: skip; the line then leads with the one-stream conv stage)")
    ap.add_argument("--high-prio", default=None,
                    help="A/B: comma-separated modalities whose backbone stream gets the high HIP priority, or 'none' "
                         "(default: the model's, Audio)")
    ap.add_argument("--stem-wgrad-last", default=None,
                    help="A/B: comma-separated modalities whose backbone issues the weight gradients of conv2_3x3 / conv2_3x3_reduce "
                         "AFTER conv1's pooled BN backward (TBN_BACKBONE_STEM_WGRAD_LAST; default: the model's -- all of them when "
                         "there are several), or 'none'")
    ap.add_argument("--no-early-flip", action="store_true",
                    help="A/B: the backbones' data-gradient weight copies at the start of each backward pass instead of right after "
                         "the modality streams are joined for the heads (TBNModel.flip_weights_early)")
    ap.add_argument("--share-stream", default=None,
                    help="A/B: 'Flow:RGB,...' -- a modality's backbone runs on another modality's stream (default: none)")
    args = ap.parse_args()
    if args.gpus > 1 and "RANK" not in os.environ and "WORLD_SIZE" not in os.environ:
        sys.exit(self_launch(args.gpus))      # the driver's 1-GPU command is plain `python bench.py`: N > 1 must work the same way
    # stdout carries exactly ONE line (the JSON result, rank 0): everything the model code prints while it builds
    # (the reference's "Freezing the batchnorms ..." notices, on every rank) goes to stderr
    # -- C libraries included (RCCL / gloo log to file descriptor 1): fd 1 is pointed at stderr for the run and the
    # JSON line is written to a duplicate of the original stdout
    sys.stdout.flush()
    json_out = os.fdopen(os.dup(1), "w")
    os.dup2(2, 1)
    sys.stdout = sys.stderr

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    smi = query_rocm_smi() if rank == 0 else None      # before anything in this process touches the GPU
    assert torch.cuda.is_available(), "bench.py needs MI355X GPUs (no CPU fallback)"
    if os.environ.get("TBN_BENCH_BACKEND", "nccl") != "nccl":
        local_rank = local_rank % torch.cuda.device_count()      # rehearsal: ranks share the card(s) present
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        # RCCL ("nccl") is the product backend; TBN_BENCH_BACKEND=gloo only exists to rehearse the N > 1 control flow
        # with several ranks sharing ONE GPU (RCCL needs a GPU per rank) -- never a measurement
        backend = os.environ.get("TBN_BENCH_BACKEND", "nccl")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=device)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
    assert world == args.gpus, (f"--gpus {args.gpus} but the launcher's WORLD_SIZE is {world}: under torch.distributed.run pass "
                                f"--nproc-per-node {args.gpus}, or run plain `python bench.py --gpus {args.gpus}` (it starts the ranks itself)")

    from attention_based_tbn_amd.config import load_config, get_modality
    from attention_based_tbn_amd.core.models import build_model
    from attention_based_tbn_amd._lib import lib
    C_ = dict(CONFIGS[args.config])
    audio_w = 256
    if args.audio_2p1s:
        assert args.config == 3, "--audio-2p1s applies to config 3"
        C_["ov"] = [o for o in C_["ov"] if not o.startswith("data.audio.audio_length")] + ["data.audio.audio_length=2.1"]
        C_["name"] = C_["name"].replace("1.279 s audio", "2.1 s audio (256x420)")
        C_["gflop"] = 110.11
        audio_w = 420
    cfg = load_config(C_["ov"])
    modality = get_modality(cfg)
    torch.manual_seed(0)
    model, criterion, _ = build_model(cfg, modality, device)
    model.train(C_["train"])
    core = getattr(model, "module", model)
    bases = [getattr(core, "Base_" + m) for m in modality]
    params = [p for p in model.parameters() if p.requires_grad]
    from attention_based_tbn_amd.core.utils import FusedSGD
    opt = FusedSGD(params, lr=cfg.train.optim.lr, momentum=cfg.train.optim.momentum,
                   weight_decay=cfg.train.optim.weight_decay)
    B = args.batch_per_gpu or C_["batch"]
    global WORKLOAD_KEY
    WORKLOAD_KEY = "config%d_B%d%s%s" % (args.config, B, "_fwd" if args.forward_only else "", "_2p1s" if args.audio_2p1s else "")
    n = cfg.train.num_segments if C_["train"] else cfg.test.num_segments
    inp, tgt = synthetic_batch(B, n, device, seed=rank, modality=modality, audio_w=audio_w)   # clips are sharded by rank: no data-path collective
    flop_per_clip = C_["gflop"] * 1e9
    if C_["train"] and args.forward_only:
        flop_per_clip = {2: 12.190, 3: 37.108 if args.audio_2p1s else 27.494, 4: 41.336}[args.config] * 1e9

    wave, spectrogram = None, None
    if args.stft_inputs:
        assert "Audio" in modality and audio_w == 256, "--stft-inputs: 1.279 s audio configs"
        from attention_based_tbn_amd.core.dataset import Spectrogram
        spectrogram = Spectrogram()
        gw = torch.Generator(device=device).manual_seed(1000 + rank)
        wave = 0.1 * torch.randn(B * n, 30695, device=device, generator=gw)      # 0.1 * N(0, 1)

    def eval_step():
        with torch.no_grad():
            out = model(inp)
        return out["verb"].float().sum() * 0 + 1.0

    def fwd_step():
        with torch.no_grad():
            out = model(inp)
            loss, _ = model.get_loss(criterion, tgt, out, 0)
        return loss["total"].detach()      # no reference to the autograd graph survives the step

    host_inp, copy_stream, staged = None, None, [None]
    if args.host_inputs:
        host_inp = {k: v.cpu().pin_memory() for k, v in inp.items()}
        copy_stream = torch.cuda.Stream(device=device)

        def stage():          # H2D of the NEXT batch on the copy stream, overlapping the current step
            with torch.cuda.stream(copy_stream):
                bufs = {k: v.to(device, non_blocking=True) for k, v in host_inp.items()}
                ev = torch.cuda.Event()
                ev.record(copy_stream)
            staged[0] = (bufs, ev)
        stage()

    def step():
        if host_inp is not None:
            bufs, ev = staged[0]
            torch.cuda.current_stream().wait_event(ev)
            for k in bufs:
                bufs[k].record_stream(torch.cuda.current_stream())
            inp.update(bufs)
            stage()
        if wave is not None:          # waveform -> (B, n, 1, 256, 256) log-power spectrogram on the GPU, every step
            inp["Audio"] = spectrogram(wave).view(B, n, 1, 256, 256)
        if not C_["train"]:
            return eval_step()
        if args.forward_only:
            return fwd_step()
        opt.zero_grad(set_to_none=True)
        out = model(inp)
        loss, _ = model.get_loss(criterion, tgt, out, 0)
        loss["total"].backward()              # N>1: gradient all-reduce is issued inside backward, waited at its end
        opt.step(clip_grad=cfg.train.clip_grad, grads_consumed=True)   # clip_grad_norm_(20) + SGD(momentum) in three HIP launches (zero_grad precedes every backward)
        return loss["total"].detach()      # no reference to the autograd graph survives the step

    buckets_last_step = None
    if world > 1 and hasattr(model, "time_sync"):
        model.time_sync = True     # HIP events around finish_gradient_sync: the all-reduce tail backward did not hide
    multi = not args.no_multi_stream
    core.multi_stream = multi
    if args.high_prio is not None:
        core.high_priority_modalities = tuple(x for x in args.high_prio.split(",") if x and x != "none")
    if args.stem_wgrad_last:
        for b_, m in zip(bases, modality):
            b_.stem_wgrad_last = m in args.stem_wgrad_last.split(",")
    if args.no_early_flip:
        core.flip_weights_early = False
    if args.share_stream:
        core.shared_streams = dict(item.split(":") for item in args.share_stream.split(","))
    aux = [b_.use_aux_stream and not args.no_aux_stream for b_ in bases]   # the model's own policy unless switched off
    if args.aux_streams is not None:
        aux = [m in args.aux_streams.split(",") for m in modality]
    for b_, a_ in zip(bases, aux):
        b_.use_aux_stream = a_
    if args.branch_streams is not None:
        want = modality if args.branch_streams == "all" else ([] if args.branch_streams == "none" else args.branch_streams.split(","))
        for b_, m in zip(bases, modality):
            b_.use_branch_streams = m in want
    branch = [b_.use_branch_streams for b_ in bases]
    if args.riders is not None:
        for b_ in bases:
            b_.use_riders = args.riders == "on"
    if args.eval_chunk > 0:
        for b_ in bases:
            b_.eval_chunk = args.eval_chunk

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    step()   # priming pass (plan creation + per-layer tile autotune happen on first use of a shape): never timed, and
             # not one of the W warm-up steps, so a small --warmup cannot push the autotuner into the timed region
    L = lib()
    L.tbn_profile_reset()
    box = box_identity(device, smi) if rank == 0 else None     # CPU / GPU names + a warm pure-MFMA burst: before the timed loop
    probe = None
    if world > 1:
        # one-shot all-reduce probe on a tensor the size of a backbone's flat weight gradient (the largest collective of a
        # step, ~41 MB), before the timed loop: makes the first real multi-GPU run self-explaining (link bandwidth vs
        # exposed_allreduce_ms).  busbw = algbw * 2 (n - 1) / n (ring all-reduce convention).
        nfl = max(int(b_.flat_weight.numel()) for b_ in bases)
        buf = torch.zeros(nfl, device=device, dtype=torch.float32)
        for _ in range(2):
            dist.all_reduce(buf)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            dist.all_reduce(buf)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 5
        t = torch.tensor([ms], device=device, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        ms = float(t.item())
        algbw = nfl * 4 / (ms * 1e-3) / 1e9
        probe = {"bytes": nfl * 4, "ms": round(ms, 3), "algbw_GBps": round(algbw, 1),
                 "busbw_GBps": round(algbw * 2 * (world - 1) / world, 1), "backend": dist.get_backend()}
        del buf
    # the W warm-up steps come LAST before the fence: the timed region then starts on a GPU that has just been running this
    # very step.  (Round 5: with the box identity / MFMA calibration / all-reduce probe between warm-up and timed loop the
    # GPU idled for tens of milliseconds first and timed step 0 was always the slowest -- 37.7 - 38.1 ms against a median of
    # 35.6, the clock ramp of hardware finding 13 -- which a 20-step run pays as 0.3 %.)
    # host hygiene first: everything built so far (modules, plans, autograd plumbing) moves to the permanent generation, so a
    # generation-2 collection inside the timed loop has little to walk (an 80-ms pause was measured in config 3)
    gc.collect()
    gc.freeze()
    for _ in range(args.warmup):
        step()
    fence()
    if world > 1 and hasattr(model, "exposed_sync_ms"):
        model.exposed_sync_ms()     # drop the warm-up records
    def profiled_step():
        # kernel timestamps on every conv-GEMM launch; the backbones run on ONE stream (and the weight gradients on it
        # too) so a bracket times exactly one kernel -- in the product step they overlap
        core.multi_stream = False
        for b_ in bases:
            b_.use_aux_stream = False
            b_.use_branch_streams = False
        L.tbn_profile_enable(1)
        out = step()
        L.tbn_profile_enable(0)
        core.multi_stream = multi
        for b_, a_, br_ in zip(bases, aux, branch):
            b_.use_aux_stream = a_
            b_.use_branch_streams = br_
        return out

    # one event per step boundary on the compute stream (the modality streams fork from and join it inside a step): the
    # per-step GPU times of the timed region, read AFTER the loop -- nothing inside the loop synchronises
    marks = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps + 1)]
    t0 = time.perf_counter()
    for i in range(args.steps):
        marks[i].record()
        if args.profile_every > 0 and (i % args.profile_every == 0):
            loss = profiled_step()      # diagnostic runs only (see --profile-every): the default timed region has none
        else:
            loss = step()
    marks[args.steps].record()
    fence()
    dt = time.perf_counter() - t0
    if world > 1 and hasattr(model, "bucket_log"):
        model.bucket_log.clear()
        step()                                  # one more (untimed) step: how many bucket collectives a step issues
        torch.cuda.synchronize()
        buckets_last_step = len(model.bucket_log)
    step_gpu_each = [marks[i].elapsed_time(marks[i + 1]) for i in range(args.steps)]
    step_gpu_ms = sorted(step_gpu_each)
    exposed_ms = None
    if world > 1 and hasattr(model, "exposed_sync_ms"):
        e = model.exposed_sync_ms()
        t = torch.tensor([e if e is not None else 0.0], device=device, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        exposed_ms = float(t.item())
    # instrumented steps AFTER the timed region (every rank runs them: the step holds collectives for N > 1); each one is
    # collected on its own so that the conv-stage figure is a median over samples, not one step
    prof_samples = []
    if args.profile_every > 0:
        prof_samples.append(collect_profile() if rank == 0 else [])
    for _ in range(max(0, args.profile_steps) if args.profile_every <= 0 else 0):
        L.tbn_profile_reset()
        profiled_step()
        torch.cuda.synchronize()
        prof_samples.append(collect_profile() if rank == 0 else [])
    if world > 1:
        fence()
    sched = None
    if args.timeline_steps > 0:
        sched = timed_schedule_conv_stage(step, args.timeline_steps, B * flop_per_clip)
        if world > 1:
            fence()
    plan_fps = {m: "+".join(sorted(b_.plan_fingerprints().values())) for m, b_ in zip(modality, bases)}
    plans_equal = None
    if world > 1:
        every_fp = [None] * world
        dist.all_gather_object(every_fp, plan_fps)
        plans_equal = all(f == every_fp[0] for f in every_fp)
    rank_ms = None
    if world > 1:
        t = torch.tensor([dt], device=device, dtype=torch.float64)
        every = [torch.zeros_like(t) for _ in range(world)]
        dist.all_gather(every, t)
        rank_ms = [1e3 * float(x.item()) / args.steps for x in every]      # each rank's own clock around the same K steps
        dt = max(float(x.item()) for x in every)                            # the job's time = the slowest rank
    assert os.environ.get("TBN_DIAG_SKIP") or torch.isfinite(loss).item(), "loss is not finite"
    if args.timeline and rank == 0:
        torch.cuda.synchronize()
        L.tbn_timeline_enable(1)
        for _ in range(4):
            step()
        torch.cuda.synchronize()
        L.tbn_timeline_enable(0)
        assert L.tbn_timeline_dump(args.timeline.encode()) == 0
    if args.trace_streams and rank == 0:
        from attention_based_tbn_amd import _lib
        for _ in range(3):
            torch.cuda.synchronize()
            _lib.TRACE = []
            ref = torch.cuda.Event(enable_timing=True)
            ref.record()
            h0 = time.perf_counter()
            step()
            h1 = time.perf_counter()
            end = torch.cuda.Event(enable_timing=True)
            end.record()
            torch.cuda.synchronize()
            tr, _lib.TRACE = _lib.TRACE, None
            print("step: GPU %.2f ms, host issue %.2f ms" % (ref.elapsed_time(end), (h1 - h0) * 1e3), file=sys.stderr)
            for name, t0, t1, e0, e1 in tr:
                print("  %-22s host %6.2f -> %6.2f ms (%.2f)   GPU %6.2f -> %6.2f ms (%.2f)" %
                      (name[4:], (t0 - h0) * 1e3, (t1 - h0) * 1e3, (t1 - t0) * 1e3, ref.elapsed_time(e0),
                       ref.elapsed_time(e1), e0.elapsed_time(e1)), file=sys.stderr)
    if rank == 0:
        clips = B * world * args.steps
        value = clips / dt
        roofline = roofline_object(prof_samples, max(1, len(range(0, args.steps, args.profile_every))) if args.profile_every > 0 else 1,
                                   value / world * flop_per_clip / 1e12 / PEAK_FP32_MFMA_TFLOPS, sched)
        line = {
            "metric": "clips/sec (3-seg RGB+Flow+Audio TBN fwd+bwd)" if args.config == 4 and not args.forward_only
            else f"clips/sec (config {args.config}{', forward only' if args.forward_only else ''})",
            "value": round(value, 2), "unit": "clips/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(1e3 * dt / args.steps, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": C_["name"] + (" [forward only]" if args.forward_only and C_["train"] else ""),
                       "batch_per_gpu": B, "global_batch": B * world, "segments": n,
                       "parallelism": f"dp{world}" if world > 1 else "single",
                       "streams": {"modality_streams": bool(multi and len(modality) > 1),
                                   "weight_gradient_stream": [m for m, a_ in zip(modality, aux) if a_],
                                   "branch_streams": [m for m, b_ in zip(modality, branch) if b_],
                                   "riders": [m for m, b_ in zip(modality, bases) if b_.use_riders],
                                   "stem_weight_gradients_last": [m for m, b_ in zip(modality, bases) if b_.stem_wgrad_last],
                                   "early_weight_flip": bool(getattr(core, "flip_weights_early", False)) and multi and len(modality) > 1},
                       **({"inputs": "pinned host memory, PCIe copy every step (diagnostic)"} if args.host_inputs else {}),
                       **({"audio_input": "waveform (30695 samples), STFT kernel inside the timed step"} if args.stft_inputs else {})},
            "roofline": roofline,
            # GPU time of each timed step (events at the step boundaries of rank 0's compute stream): a clock ramp or a
            # straggler step shows up here, not in the mean
            "step_gpu_ms": {"median": round(step_gpu_ms[len(step_gpu_ms) // 2], 3), "min": round(step_gpu_ms[0], 3),
                            "max": round(step_gpu_ms[-1], 3), "max_at_step": step_gpu_each.index(step_gpu_ms[-1]),
                            "slowest_five": [[i, round(t, 3)] for t, i in sorted(((t, i) for i, t in enumerate(step_gpu_each)),
                                                                                  reverse=True)[:5]]},
        }
        if world > 1:
            line["multi_gpu"] = {"dist_world_size": dist.get_world_size(), "cuda_device_count": torch.cuda.device_count(),
                                 # collectives issued from INSIDE the backbones' backward passes in the last step (two per
                                 # backbone: inception_5a..5b, then 4a..4e; the prefix follows from the gradient hook)
                                 "gradient_buckets_per_step": buckets_last_step,
                                 "ms_per_step_rank_min": round(min(rank_ms), 3), "ms_per_step_rank_max": round(max(rank_ms), 3),
                                 "allreduce_probe": probe}
        if exposed_ms is not None:
            # mean GPU time per step between the end of the last backbone's backward (entry of the gradient-sync
            # callback on the compute stream) and the return of finish_gradient_sync, max over ranks
            line["exposed_allreduce_ms"] = round(exposed_ms, 3)
        # fingerprint of each backbone's tuned launch plan (tbn_backbone_plan_fingerprint): equal lines on two boxes with
        # different numbers = a slow box; different fingerprints = a different plan.  N > 1: rank 0's plans are
        # broadcast at first use (DataParallel / PlanSync), `plans_equal_across_ranks` confirms it
        box["plans"] = plan_fps
        if world > 1:
            line["multi_gpu"]["plans_equal_across_ranks"] = plans_equal
        line["box"] = box
        if args.profile_every > 0:
            line["timed_region_instrumented"] = True      # diagnostic run: `value` includes single-stream instrumented steps
        if world == 1 and not args.no_cpu_baseline and args.config == 4 and not args.forward_only:
            line["cpu_baseline"] = cpu_baseline()
        print(json.dumps(line), file=json_out, flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
