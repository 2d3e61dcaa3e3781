#!/bin/bash
# VALU instruction counts of the tile-scaled int8 kernel (DeepSeek-style blocks, 4096^3, A@B.T) with the exact three-rounding rescale and with
# the fused one: tools/pmc_tile_valu.sh <tag>   (GPU box, repo root; one counter group per run)
set -e
tag=$1
repo=$PWD
cd /tmp && export TMPDIR=/tmp
for mode in exact fused; do
  out=$repo/gpurun_out/pmc_${tag}_$mode
  mkdir -p $out
  export GNK_TILE_RESCALE=$mode
  timeout -s KILL 120 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY -d $out/sq -o sq --output-format csv -- python3 $repo/tools/one_gemm.py i8 nt 4096 20 -1 128 > $out/run.log 2>&1
  echo "## rescale = $mode"
  python3 $repo/tools/pmc_summary.py $out
done
