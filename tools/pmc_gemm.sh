#!/bin/bash
# PMC passes over one GEMM configuration (run on the GPU box from the repo root): tools/pmc_gemm.sh <tag> <one_gemm.py args...>
# One counter group per run (rocprofv3 does not split counters over passes).
set -e
tag=$1; shift
out=$PWD/gpurun_out/pmc_$tag
mkdir -p $out
repo=$PWD
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 120 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE -d $out/sq -o sq --output-format csv -- python3 $repo/tools/one_gemm.py "$@"
timeout -s KILL 120 rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS -d $out/lds -o lds --output-format csv -- python3 $repo/tools/one_gemm.py "$@"
timeout -s KILL 120 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $out/fetch -o fetch --output-format csv -- python3 $repo/tools/one_gemm.py "$@"
timeout -s KILL 120 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $out/write -o write --output-format csv -- python3 $repo/tools/one_gemm.py "$@"
find $out -name "*.csv" | head -40
