#!/bin/bash
# Output-store flavours of the one-wave-per-SIMD int8 GEMM (CPU side): builds libgnk_st<N>.so with -DGNK_STORE_MODE=N (gnk_common.h).
# Run on the GPU box afterwards: GNK_CMP_VARIANTS=w4 tools/ab_libs.sh "tools/cmp_variants.py 4096" "" libgnk_st1.so ...
set -e
cd "$(dirname "$0")/../quantized-training_amd/csrc"
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -w"
mkdir -p build_abl
for n in "$@"; do
  /opt/rocm/bin/hipcc $FLAGS -DGNK_STORE_MODE=$n -c gnk_gemm_i8_v5.hip -o build_abl/gnk_gemm_i8_v5_st$n.o &
  /opt/rocm/bin/hipcc $FLAGS -DGNK_STORE_MODE=$n -c gnk_gemm_f8_00_v5.hip -o build_abl/gnk_gemm_f8_00_v5_st$n.o &
  wait
  rest=$(ls build/*.o | grep -v "gnk_gemm_i8_v5.o\|gnk_gemm_f8_00_v5.o")
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../gn_kernels/libgnk_st$n.so $rest build_abl/gnk_gemm_i8_v5_st$n.o build_abl/gnk_gemm_f8_00_v5_st$n.o
done
