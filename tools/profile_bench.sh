#!/bin/bash
# rocprofv3 kernel stats of the bench command (GPU box, from the repo root): tools/profile_bench.sh <tag>
set -e
tag=$1
repo=$PWD
out=$repo/gpurun_out/prof_$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $out -o bench --output-format csv -- python3 $repo/bench.py --no-cpu --no-train --no-detail > $out/bench_under_rocprof.json 2> $out/rocprof.log
find $out -name "*.csv"
