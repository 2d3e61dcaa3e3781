#!/bin/bash
# A/B builds of the quantize kernels (CPU side): tools/quant_ab.sh <tag> "<-D flags>" ... -> gn_kernels/libgnk_q<tag>.so
# Run on the GPU box: GNK_NO_TORCH_SHIM=1 tools/ab_libs.sh tools/quant_step_probe.py "" libgnk_q<tag>.so ...
set -e
cd "$(dirname "$0")/../quantized-training_amd/csrc"
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -w"
mkdir -p build_abl
while [ $# -ge 2 ]; do
  tag=$1; defs=$2; shift 2
  /opt/rocm/bin/hipcc $FLAGS $defs -c gnk_quantize.hip -o build_abl/gnk_quantize_$tag.o
  rest=$(ls build/*.o | grep -v "gnk_quantize.o")
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../gn_kernels/libgnk_q$tag.so $rest build_abl/gnk_quantize_$tag.o
done
