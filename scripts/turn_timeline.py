"""The serial 'turn' of a training step between the last backbone forward and the first backbone backward GEMM (heads,
loss, their backward, the backward prologues), from a `rocprofv3 --kernel-trace` CSV of
`bench.py --steps 6 --warmup 2 --no-cpu-baseline --profile-steps 0`: every kernel with start / duration relative to the end
of the last conv GEMM of the forward passes of the LAST traced step."""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
names = [r["Kernel_Name"] for r in rows]
# last optimiser step = end of the last step; walk back to the previous one = start of that step
sgd = [i for i, n in enumerate(names) if "opt_sgd_kernel" in n]
a, b = sgd[-2], sgd[-1]
step = rows[a + 1:b + 1]
t0 = int(step[0]["Start_Timestamp"])
fwd_last = max(i for i, r in enumerate(step) if "spatial_mean_fwd" in r["Kernel_Name"])
flip_first = min(i for i, r in enumerate(step) if "weight_flip" in r["Kernel_Name"])
gemm_first = min(i for i, r in enumerate(step) if i > flip_first and ("conv_" in r["Kernel_Name"]))
ref = int(step[fwd_last]["End_Timestamp"])
print("step %.3f ms; turn (last spatial_mean_fwd end -> first backward conv kernel start) %.1f us, %d kernels" %
      ((int(step[-1]["End_Timestamp"]) - t0) / 1e6, (int(step[gemm_first]["Start_Timestamp"]) - ref) / 1e3, gemm_first - fwd_last - 1))
for r in step[fwd_last - 3:gemm_first + 1]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print("%9.1f us  +%7.1f us  %s" % ((s - ref) / 1e3, (e - s) / 1e3, r["Kernel_Name"][:110]))
tail = [r for r in step if "opt_" in r["Kernel_Name"]]
print("optimiser: %.1f us from the first opt kernel's start to the last one's end" %
      ((int(tail[-1]["End_Timestamp"]) - int(tail[0]["Start_Timestamp"])) / 1e3))
