"""Builds libtbn_hip.so (gfx950) in-tree with hipcc.  `python -m attention_based_tbn_amd.build`."""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libtbn_hip.so")
SOURCES = ["api.hip", "conv_igemm.hip", "bn.hip", "bn_multi.hip", "pool.hip", "heads.hip", "stft.hip", "engine.hip", "train_ops.hip", "frames.hip"]
# -amdgpu-mfma-vgpr-form: the MFMA accumulators live in VGPRs instead of AGPRs.  With AGPR accumulators the compiler
# copied every accumulator register AGPR -> VGPR -> AGPR once per K-loop iteration around some loop shapes (the split-K
# tile kernel: 32 copies per 16 MFMAs; the LDS-DMA and LDS-halo kernels likewise) -- VALU instructions on the port the
# fp32 MFMA shares, plus an MFMA pipeline drain in front of each copy block -- and rounded the VGPR / AGPR halves up
# separately (scripts/isa_scan.py: fewer registers for most instantiations, one more wave per SIMD for a dozen of them).
FLAGS = ["--offload-arch=gfx950", "-O3", "-fPIC", "-std=c++17", "-Wall", "-Wno-unused-function",
         "-mllvm", "-amdgpu-mfma-vgpr-form",
         "-I" + os.path.join(os.path.dirname(HERE), "include")]
if os.environ.get("TBN_DIAG") == "1":         # timing-diagnostic build (engine can skip kernel groups; results invalid)
    FLAGS.append("-DTBN_DIAG=1")
if os.environ.get("TBN_EXTRA_FLAGS"):         # experiments (same-box A/B of a -D switch): never the shipped build
    FLAGS += os.environ["TBN_EXTRA_FLAGS"].split()
if os.environ.get("TBN_ABLATE") == "1":      # timing-ablation build for scripts/conv_ablate.py (never the shipped one)
    FLAGS.append("-DTBN_ABLATE=1")
if os.environ.get("TBN_EXPERIMENT") == "1":  # A/B build: the only one whose library reads experiment knobs from the environment
    FLAGS.append("-DTBN_EXPERIMENT=1")       # (tbn_common.h: tbn_env_int; the shipped library reads no environment variable)


def _hipcc():
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if c and (os.path.isabs(c) and os.path.exists(c) or not os.path.isabs(c)):
            return c
    raise RuntimeError("hipcc not found")


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


# Diagnostic / experiment builds (TBN_DIAG=1, TBN_ABLATE=1, TBN_EXTRA_FLAGS) never replace the shipped library: with
# TBN_BUILD_VARIANT=<name> objects and library go to scripts/ab/obj_<name>/ and scripts/ab/lib_<name>.so, and a run picks the
# variant with TBN_LIB=<path> (attention_based_tbn_amd/_lib.py) -- nothing is copied over libtbn_hip.so (round-4 advisor)
VARIANT = os.environ.get("TBN_BUILD_VARIANT")
if not VARIANT and any(os.environ.get(k) == "1" for k in ("TBN_EXPERIMENT", "TBN_DIAG", "TBN_ABLATE")):
    VARIANT = "exp" if os.environ.get("TBN_EXPERIMENT") == "1" else ("diag" if os.environ.get("TBN_DIAG") == "1" else "ablate")
OBJDIR = CSRC
if VARIANT:
    _ab = os.path.join(os.path.dirname(HERE), "scripts", "ab")
    OBJDIR = os.path.join(_ab, "obj_" + VARIANT)
    LIB = os.path.join(_ab, "lib_" + VARIANT + ".so")
    os.makedirs(OBJDIR, exist_ok=True)
STAMP = os.path.join(OBJDIR, ".build_flags")   # the flag set the objects were built with (diagnostic -D switches included)


def build(force=False, verbose=True):
    hipcc = _hipcc()
    flagset = " ".join(FLAGS)
    try:
        with open(STAMP) as f:
            if f.read() != flagset:
                force = True         # e.g. a TBN_DIAG / TBN_ABLATE build before: never link those objects into a plain build
    except OSError:
        force = force or any(os.path.exists(os.path.join(OBJDIR, s.replace(".hip", ".o"))) for s in SOURCES)
    headers = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")]
    headers.append(os.path.join(os.path.dirname(HERE), "include", "tbn_hip.h"))
    objs, jobs = [], []
    for s in SOURCES:
        src = os.path.join(CSRC, s)
        obj = os.path.join(OBJDIR, s.replace(".hip", ".o"))
        objs.append(obj)
        if force or _stale(obj, [src] + headers):
            jobs.append([hipcc] + FLAGS + ["-c", src, "-o", obj])

    def run(cmd):
        if verbose:
            print(" ".join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("hipcc failed:\n" + r.stdout + r.stderr)
        if verbose and r.stderr.strip():
            print(r.stderr, file=sys.stderr)

    with ThreadPoolExecutor(max_workers=min(8, max(1, len(jobs)))) as ex:
        list(ex.map(run, jobs))
    with open(STAMP, "w") as f:
        f.write(flagset)
    if jobs or force or _stale(LIB, objs):
        run([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))
