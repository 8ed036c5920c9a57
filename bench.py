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
        k = doc["kernels"].get(kernel)
        # `read` carries the guide's gfx950 correction (FETCH_SIZE x 2: it counts a 128-B request as 64 B): exact for wide
        # coalesced reads, an UPPER bound for kernels whose requests are 64 B wide (the weight-gradient kernel: four lanes x
        # 16 B per pixel row) -- `read_uncorrected` is the matching lower bound
        return None if k is None else {"hbm_bytes_per_launch": k["hbm_bytes_per_launch"],
                                        "read": k["hbm_read_bytes_per_launch"], "write": k["hbm_write_bytes_per_launch"],
                                        "read_uncorrected": round(1024.0 * k["fetch_kib_raw_per_launch"]),
                                        "source": src}
    except (OSError, ValueError, KeyError, ImportError):
        return None


def cpu_baseline(seconds_budget=25.0):
    """the CPU oracle (torch-CPU restatement of the reference path) on this box's host cores:
    same config-4 graph and full-size inputs, B=2 clips x 3 segments, fwd+loss+bwd"""
    from attention_based_tbn_amd.config import load_config, get_modality
    from oracle.fill import pretrained_pair
    from oracle.tbn import build_model as build_oracle
    cfg = load_config(["model.attention.enable=False", "data.audio.audio_length=1.279", "data.sampling=async"])
    modality = get_modality(cfg)
    torch.manual_seed(0)
    model, crit, _ = build_oracle(cfg, modality, pretrained_pair(7))
    model.train()
    B, n = 2, 3
    g = torch.Generator().manual_seed(0)
    inp = {"RGB": torch.rand(B, n, 3, 224, 224, generator=g) - 0.45,
           "Flow": torch.rand(B, n, 10, 224, 224, generator=g) - 0.5,
           "Audio": torch.randn(B, n, 1, 256, 256, generator=g) * 3 - 6}
    tgt = {"class": {"verb": torch.randint(0, 125, (B,), generator=g), "noun": torch.randint(0, 352, (B,), generator=g)}}

    def step():
        model.zero_grad()
        out = model(inp)
        loss, _ = model.get_loss(crit, tgt, out, 0)
        loss["total"].backward()

    step()  # warm-up
    t0 = time.perf_counter()
    steps = 0
    while steps < 2 or (time.perf_counter() - t0 < seconds_budget and steps < 8):
        step()
        steps += 1
    dt = time.perf_counter() - t0
    return {"value": B * steps / dt, "unit": "clips/s", "cores": torch.get_num_threads(), "kind": "port",
            "sample": f"{steps} fwd+bwd steps of the config-4 graph at B={B} clips x {n} segments, full-size "
                      f"synthetic inputs, oracle/ (torch-CPU fp32, oneDNN) on the GPU box's host cores"}


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
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--host-inputs", action="store_true",
                    help="diagnostic (never the headline): the batch starts in pinned HOST memory every step and is copied "
                         "over PCIe on a side stream, double-buffered against the previous step -- the PCIe-inclusive rate")
    ap.add_argument("--stft-inputs", action="store_true",
                    help="the audio leg starts from WAVEFORMS (B, n, 30695 samples = 1.279 s at 24 kHz, resident in HBM): the "
                         "log-power STFT kernel (reference dataset.py:483-495, CPU librosa there) runs inside the timed step")
    ap.add_argument("--profile-every", type=int, default=1 << 30,
                    help="bracket conv-GEMM launches with HIP events on every k-th timed step (default: the first "
                         "timed step only; 0 = never).  Profiled steps run single-stream, see DESIGN.md")
    ap.add_argument("--trace-streams", action="store_true",
                    help="diagnostic: after the timed loop run 3 more steps with HIP events around every backbone "
                         "forward / backward call and print (stderr) when each ran on the GPU and how long the host took "
                         "to issue it -- the stagger between the modality streams")
    args = ap.parse_args()
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
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run"

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
        opt.step(clip_grad=cfg.train.clip_grad)   # clip_grad_norm_(20) + SGD(momentum) in three HIP launches
        return loss["total"].detach()      # no reference to the autograd graph survives the step

    if world > 1 and hasattr(model, "time_sync"):
        model.time_sync = True     # HIP events around finish_gradient_sync: the all-reduce tail backward did not hide
    multi = not args.no_multi_stream
    core.multi_stream = multi
    aux = [b_.use_aux_stream and not args.no_aux_stream for b_ in bases]   # the model's own policy unless switched off
    if args.aux_streams is not None:
        aux = [m in args.aux_streams.split(",") for m in modality]
    for b_, a_ in zip(bases, aux):
        b_.use_aux_stream = a_

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    step()   # priming pass (plan creation + per-layer tile autotune happen on first use of a shape): never timed, and
             # not one of the W warm-up steps, so a small --warmup cannot push the autotuner into the timed region
    for _ in range(args.warmup):
        step()
    L = lib()
    L.tbn_profile_reset()
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
    fence()
    if world > 1 and hasattr(model, "exposed_sync_ms"):
        model.exposed_sync_ms()     # drop the warm-up records
    t0 = time.perf_counter()
    for i in range(args.steps):
        prof = args.profile_every > 0 and (i % args.profile_every == 0)
        if prof:
            # HIP-event brackets around every conv-GEMM launch; on these steps the three backbones run
            # on ONE stream so a bracket times exactly one kernel (on the other steps they overlap)
            core.multi_stream = False
            for b_ in bases:
                b_.use_aux_stream = False
            L.tbn_profile_enable(1)
        loss = step()
        if prof:
            L.tbn_profile_enable(0)
            core.multi_stream = multi
            for b_, a_ in zip(bases, aux):
                b_.use_aux_stream = a_
    fence()
    dt = time.perf_counter() - t0
    rank_ms = None
    if world > 1:
        t = torch.tensor([dt], device=device, dtype=torch.float64)
        every = [torch.zeros_like(t) for _ in range(world)]
        dist.all_gather(every, t)
        rank_ms = [1e3 * float(x.item()) / args.steps for x in every]      # each rank's own clock around the same K steps
        dt = max(float(x.item()) for x in every)                            # the job's time = the slowest rank
    assert os.environ.get("TBN_DIAG_SKIP") or torch.isfinite(loss).item(), "loss is not finite"
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
    exposed_ms = None
    if world > 1 and hasattr(model, "exposed_sync_ms"):
        e = model.exposed_sync_ms()
        t = torch.tensor([e if e is not None else 0.0], device=device, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        exposed_ms = float(t.item())

    if rank == 0:
        clips = B * world * args.steps
        value = clips / dt
        prof_all = sorted(collect_profile(), key=lambda e: -e["ms"])
        # the BN-Inception conv stage (what the north star's roofline target is about) vs the head Linear GEMMs
        # (fusion / classifier / attention projections: M = 32 ... 96 rows, < 0.2 % of the FLOPs, latency-bound)
        prof = [e for e in prof_all if not e["kernel"].startswith("linear: ")]
        heads = [e for e in prof_all if e["kernel"].startswith("linear: ")]
        roofline = None
        if prof:
            top = prof[0]
            ach = top["flops"] / (top["ms"] * 1e-3) / 1e12
            tot_ms, tot_fl = sum(e["ms"] for e in prof), sum(e["flops"] for e in prof)
            tr = pmc_traffic(top["kernel"])
            tr_detail = tr
            if tr and tr.get("stale"):
                tr = None                      # stale counters: traffic is null, the detail says why
            alg_b = top["alg_bytes"] / max(1, top["launches"])
            fam = {}
            for e in prof:                     # launches per kernel family in the profiled step(s): which variants the
                f_ = e["kernel"].split("<")[0]  # per-layer autotuner actually put on this shape
                fam[f_] = fam.get(f_, 0) + e["launches"]
            roofline = {"bound": "mfma", "achieved": round(ach, 2), "peak": PEAK_FP32_MFMA_TFLOPS, "unit": "TFLOP/s",
                        "frac": round(ach / PEAK_FP32_MFMA_TFLOPS, 4), "traffic": (tr or {}).get("hbm_bytes_per_launch"),
                        "alg_bytes_per_launch": round(alg_b),
                        "traffic_ratio": round(tr["hbm_bytes_per_launch"] / alg_b, 3) if tr and alg_b > 0 else None,
                        "traffic_detail": tr_detail,
                        "kernel_families": fam,
                        "kernel": top["kernel"], "launches": top["launches"],
                        "avg_launch_us": round(1e3 * top["ms"] / top["launches"], 2),
                        "alg_gflop_per_launch": round(top["flops"] / top["launches"] / 1e9, 4),
                        "all_conv_gemm": {"achieved": round(tot_fl / (tot_ms * 1e-3) / 1e12, 2),
                                          "frac": round(tot_fl / (tot_ms * 1e-3) / 1e12 / PEAK_FP32_MFMA_TFLOPS, 4),
                                          "ms_per_profiled_step": round(tot_ms / max(1, len(range(0, args.steps, args.profile_every))), 2)},
                        "head_linear_gemm": {"launches": sum(e["launches"] for e in heads),
                                             "ms_per_profiled_step": round(sum(e["ms"] for e in heads) /
                                                                           max(1, len(range(0, args.steps, args.profile_every))), 3)},
                        "end_to_end_frac": round(value / world * flop_per_clip / 1e12 / PEAK_FP32_MFMA_TFLOPS, 4),
                        "by_kernel": [{"kernel": e["kernel"], "launches": e["launches"],
                                       "avg_us": round(1e3 * e["ms"] / e["launches"], 2),
                                       "tflops": round(e["flops"] / (e["ms"] * 1e-3) / 1e12, 2)} for e in prof[:8]]}
        line = {
            "metric": "clips/sec (3-seg RGB+Flow+Audio TBN fwd+bwd)" if args.config == 4 and not args.forward_only
            else f"clips/sec (config {args.config}{', forward only' if args.forward_only else ''})",
            "value": round(value, 2), "unit": "clips/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(1e3 * dt / args.steps, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": C_["name"] + (" [forward only]" if args.forward_only and C_["train"] else ""),
                       "batch_per_gpu": B, "global_batch": B * world, "segments": n,
                       "parallelism": f"dp{world}" if world > 1 else "single",
                       **({"inputs": "pinned host memory, PCIe copy every step (diagnostic)"} if args.host_inputs else {}),
                       **({"audio_input": "waveform (30695 samples), STFT kernel inside the timed step"} if args.stft_inputs else {})},
            "roofline": roofline,
        }
        if world > 1:
            line["multi_gpu"] = {"dist_world_size": dist.get_world_size(), "cuda_device_count": torch.cuda.device_count(),
                                 "ms_per_step_rank_min": round(min(rank_ms), 3), "ms_per_step_rank_max": round(max(rank_ms), 3),
                                 "allreduce_probe": probe}
        if exposed_ms is not None:
            # mean GPU time per step between the end of the last backbone's backward (entry of the gradient-sync
            # callback on the compute stream) and the return of finish_gradient_sync, max over ranks
            line["exposed_allreduce_ms"] = round(exposed_ms, 3)
        if world == 1 and not args.no_cpu_baseline and args.config == 4 and not args.forward_only:
            line["cpu_baseline"] = cpu_baseline()
        print(json.dumps(line), file=json_out, flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
