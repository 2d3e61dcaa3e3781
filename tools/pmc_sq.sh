#!/bin/bash
# SQ pass only: tools/pmc_sq.sh <tag> <one_gemm.py args...>
set -e
tag=$1; shift
out=$PWD/gpurun_out/pmc_$tag
mkdir -p $out
repo=$PWD
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 120 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE -d $out/sq -o sq --output-format csv -- python3 $repo/tools/one_gemm.py "$@" > $out/sq.log 2>&1
