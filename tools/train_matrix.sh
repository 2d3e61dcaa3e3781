#!/bin/bash
# Train-bench matrix for profiles/rNN_train_bench.md: 470m and "1b", bf16 vs INT8-MP, eager vs torch.compile.  One GPU.
# usage: tools/train_matrix.sh OUT.jsonl
out=${1:-gpurun_out/train_matrix.jsonl}
: > "$out"
run() { echo "# $*" >> "$out"; timeout -k 10 400 python -u train_bench.py "$@" 2>>"${out%.jsonl}.err" | grep '^{' >> "$out" || { echo "FAILED: $*" >> "$out"; return 1; }; }
run --model 470m --quantize none --steps 10 --warmup 3 &&
run --model 470m --quantize mixed_precision --steps 10 --warmup 3 &&
run --model 470m --quantize none --steps 10 --warmup 3 --compile &&
run --model 470m --quantize mixed_precision --steps 10 --warmup 3 --compile &&
run --model 1b --batch_size 16 --quantize none --steps 6 --warmup 2 &&
run --model 1b --batch_size 16 --quantize mixed_precision --steps 6 --warmup 2 &&
run --model 1b --batch_size 16 --quantize none --steps 6 --warmup 2 --compile &&
run --model 1b --batch_size 16 --quantize mixed_precision --steps 6 --warmup 2 --compile
