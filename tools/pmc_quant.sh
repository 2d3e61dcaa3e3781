#!/bin/bash
# HBM-side PMC passes over the quantize kernels (GPU box, repo root): tools/pmc_quant.sh <tag> <R> <C> <dim>
# FETCH_SIZE and WRITE_SIZE in separate runs (rocprofv3 does not split counters over passes).
set -e
tag=$1; shift
out=$PWD/gpurun_out/pmcq_$tag
mkdir -p $out
repo=$PWD
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 120 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $out/fetch -o fetch --output-format csv -- python3 $repo/tools/one_quant.py "$@"
timeout -s KILL 120 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $out/write -o write --output-format csv -- python3 $repo/tools/one_quant.py "$@"
