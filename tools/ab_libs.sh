#!/bin/bash
# A/B of library builds on ONE box: tools/ab_libs.sh "<python tool + args>" lib1.so lib2.so ...   ("" = the shipped library)
cmd=$1; shift
for lib in "$@"; do
  echo "== ${lib:-shipped}"
  if [ -n "$lib" ]; then export GNK_LIBRARY=$PWD/quantized-training_amd/gn_kernels/$lib; else unset GNK_LIBRARY; fi
  timeout -k 10 300 python -u $cmd || exit 1
done
