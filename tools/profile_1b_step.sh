#!/bin/bash
# rocprofv3 kernel trace of the compiled Llama-1B INT8 mixed-precision step + per-category summary: tools/profile_1b_step.sh <tag>
set -e
tag=${1:-1b}
repo=$PWD
out=$repo/gpurun_out/prof_$tag; mkdir -p $out
(cd /tmp && TMPDIR=/tmp timeout -k 10 400 rocprofv3 --kernel-trace --stats -d $out -o run --output-format csv -- python3 $repo/train_bench.py --model 1b --batch_size 16 --quantize mixed_precision --compile --steps 5 --warmup 2 > $out/stdout.log 2> $out/rocprof.log)
tail -1 $out/stdout.log
python3 tools/train_profile_summary.py $out/run_kernel_trace.csv 2 > $out/summary.txt
head -32 $out/summary.txt
