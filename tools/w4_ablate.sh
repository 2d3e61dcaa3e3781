#!/bin/bash
# Timing ablations of the one-wave-per-SIMD GEMM loop (CPU side): builds libgnk_w4abl<N>.so for each flag set N given
# (tools/gen_w4_asm.py: 1 no DMA, 2 no fragment reads, 4 no barrier, 8 no MFMAs -- results WRONG by construction; 16 clock stamps,
# results right, read back with gnk_debug_w4_stamps: tools/w4_clock.py).
# Run on the GPU box afterwards: tools/ab_libs.sh "tools/cmp_variants.py 4096" "" libgnk_w4abl1.so ...
set -e
cd "$(dirname "$0")/../quantized-training_amd/csrc"
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -w"
mkdir -p build_abl
for n in "$@"; do
  GNK_W4_ABL=$n python3 ../../tools/gen_w4_asm.py >/dev/null
  objs=""
  extra=""
  if [ $((n & 16)) -ne 0 ]; then extra="-DGNK_W4_STAMP"; fi
  for u in gnk_gemm_i8_v5 gnk_gemm_f8_00_v5; do
    /opt/rocm/bin/hipcc $FLAGS $extra -c $u.hip -o build_abl/${u}_$n.o &
    objs="$objs build_abl/${u}_$n.o"
  done
  wait
  rest=$(ls build/*.o | grep -v "gnk_gemm_i8_v5.o\|gnk_gemm_f8_00_v5.o")
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../gn_kernels/libgnk_w4abl$n.so $rest $objs
done
python3 ../../tools/gen_w4_asm.py >/dev/null
