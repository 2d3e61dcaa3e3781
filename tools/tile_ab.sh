#!/bin/bash
# tile-scaled kernels, two library builds on one box, two rounds: tools/tile_ab.sh libA.so libB.so   ("" = shipped)
for round in 1 2; do
  bash tools/ab_libs.sh "tools/tile_rates.py" "$@" || exit 1
done
