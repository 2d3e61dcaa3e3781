#!/bin/bash
# rocprofv3 kernel stats of any python tool: tools/profile_cmd.sh <tag> <script.py> [args...]
set -e
tag=$1; shift
script=$1; shift
repo=$PWD
out=$repo/gpurun_out/prof_$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $out -o run --output-format csv -- python3 $repo/$script "$@" > $out/stdout.log 2> $out/rocprof.log
cat $out/stdout.log
cut -d, -f1-4 $out/run_kernel_stats.csv | cut -c1-150
