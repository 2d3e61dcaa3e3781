"""Per-category kernel time of one profiled train_bench run: tools/train_profile_summary.py <run_kernel_trace.csv> <steps_profiled>
Groups the rocprofv3 kernel trace into the step's parts (gn_kernels GEMMs, quantize passes, attention, vendor GEMMs, compiler-generated
or ATen glue, optimizer) and prints ms per training step."""
import collections
import csv
import sys

rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r["Start_Timestamp"]))
steps = int(sys.argv[2])      # how many of the LAST training steps to average over (steady state: compile / autotune kernels come first)
# a training step ends with the optimizer's multi_tensor_apply launches: cut the trace after each such run
ends = [i for i, r in enumerate(rows) if "multi_tensor_apply" in r["Kernel_Name"] and (i + 1 == len(rows) or "multi_tensor_apply" not in rows[i + 1]["Kernel_Name"])]
# the optimizer runs several multi_tensor groups per step with other kernels between them: keep only ends followed by a non-optimizer stretch
cuts = [e for k, e in enumerate(ends) if k + 1 == len(ends) or ends[k + 1] - e > 200]
assert len(cuts) > steps, f"only {len(cuts)} optimizer steps in the trace"
rows = rows[cuts[-steps - 1] + 1:cuts[-1] + 1]
wall = (int(rows[-1]["End_Timestamp"]) - int(rows[0]["Start_Timestamp"])) / 1e6
print(f"steady state: last {steps} steps, {len(rows)} launches, wall {wall / steps:.1f} ms/step")


def cat(n):
    if "gnk_gemm" in n or "splitk_reduce" in n:
        return "gn_kernels GEMM"
    if "quant_" in n or "col_absmax" in n or "col_quant" in n or "colmax_reduce" in n:
        return "gn_kernels quantize"
    if "attn" in n.lower() or "fmha" in n.lower() or "flash" in n.lower() or "aotriton" in n.lower() or n.startswith("bwd_kernel") or n.startswith("bwd_preprocess"):
        return "attention (SDPA)"
    if n.startswith("Cijk") or "gemm" in n.lower() or "rocblas" in n.lower() or "hipblas" in n.lower():
        return "vendor GEMM (hipBLASLt: lm_head, bf16 Linears)"
    if "triton" in n:
        return "inductor (triton) glue"
    if "multi_tensor" in n or "adam" in n.lower():
        return "optimizer"
    if "fill" in n.lower() or "memset" in n.lower():
        return "fills"
    return "ATen / other"


tot = collections.defaultdict(float)
cnt = collections.defaultdict(int)
per = collections.defaultdict(float)
for r in rows:
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
    c = cat(r["Kernel_Name"])
    tot[c] += d
    cnt[c] += 1
    per[(c, r["Kernel_Name"][:90])] += d
all_ms = sum(tot.values())
print(f"sum of kernel durations: {all_ms / steps:.1f} ms/step")
for c, v in sorted(tot.items(), key=lambda kv: -kv[1]):
    print(f"  {c:50s} {v / steps:8.2f} ms/step  {cnt[c] / steps:8.0f} launches/step")
print("top kernels:")
for (c, n), v in sorted(per.items(), key=lambda kv: -kv[1])[:25]:
    print(f"  {v / steps:8.2f} ms/step  [{c[:12]}] {n}")
