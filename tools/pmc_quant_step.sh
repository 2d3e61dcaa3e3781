#!/bin/bash
# SQ / memory-side PMC passes over the in-step quantize kernels (GPU box, repo root): tools/pmc_quant_step.sh <tag> <R> <C>
set -e
tag=$1; shift
out=$PWD/gpurun_out/pmcqs_$tag
mkdir -p $out
repo=$PWD
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 120 rocprofv3 --kernel-trace --pmc SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU -d $out/sq -o sq --output-format csv -- python3 $repo/tools/one_quant_step.py "$@"
timeout -s KILL 120 rocprofv3 --kernel-trace --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_WAVES -d $out/sq2 -o sq2 --output-format csv -- python3 $repo/tools/one_quant_step.py "$@"
timeout -s KILL 120 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $out/fetch -o fetch --output-format csv -- python3 $repo/tools/one_quant_step.py "$@"
timeout -s KILL 120 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $out/write -o write --output-format csv -- python3 $repo/tools/one_quant_step.py "$@"
