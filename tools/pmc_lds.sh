#!/bin/bash
# LDS counters only: tools/pmc_lds.sh <tag> <one_gemm.py args...>
set -e
tag=$1; shift
out=$PWD/gpurun_out/pmc_$tag
mkdir -p $out
repo=$PWD
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 120 rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL -d $out/lds -o lds --output-format csv -- python3 $repo/tools/one_gemm.py "$@" > $out/lds.log 2>&1
