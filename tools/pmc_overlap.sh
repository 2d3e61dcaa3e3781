#!/bin/bash
# MFMA / VALU co-execution probe under counters: tools/probe_overlap (SLP-packed v_pk_* rescale) and
# tools/probe_overlap_scalar (-fno-slp-vectorize: scalar v_mul_f32 / v_add_f32), plain timing then one PMC pass each.
out=$PWD/gpurun_out/pmc_overlap
mkdir -p $out
repo=$PWD
cd /tmp && export TMPDIR=/tmp
for b in probe_overlap probe_overlap_scalar; do
  timeout -k 10 120 $repo/tools/$b > $out/$b.txt 2>&1 || exit 1
  timeout -s KILL 120 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE -d $out/$b -o pmc --output-format csv -- $repo/tools/$b > $out/$b.pmc.log 2>&1 || exit 1
done
rocprofv3 -L 2>/dev/null | grep -i -A2 "COEXEC\|MFMA" | head -40 > $out/counters_mfma.txt
