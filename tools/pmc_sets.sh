#!/bin/bash
# usage: tools/pmc_sets.sh <kernel-filter> <outfile> -- <command...>   (several --pmc passes, one summary line per set)
# Counters are collected in their own runs, with --kernel-trace only (never with sys / hip / hsa tracing).
flt=$1; out=$2; shift 3
export TMPDIR=/tmp
R=$PWD
: > $out
while read -r set; do
  [ -z "$set" ] && continue
  rm -rf /tmp/pmcset
  ( cd /tmp && rocprofv3 --kernel-trace --pmc $set -d /tmp/pmcset -o p --output-format csv -- "$@" > /tmp/pmcset.log 2>&1 )
  python $R/tools/pmc_summary.py /tmp/pmcset "$flt" >> $out 2>&1 || tail -3 /tmp/pmcset.log >> $out
done <<SETS
SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY
SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE SQ_ACTIVE_INST_VALU
SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_LDS SQ_INSTS_VALU SQ_WAIT_INST_LDS
SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_LDS_BANK_CONFLICT
FETCH_SIZE
WRITE_SIZE
SETS
cat $out
