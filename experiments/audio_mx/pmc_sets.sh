#!/bin/bash
# usage: tools/pmc_sets.sh <kernel-filter> <outfile> -- <command...>   (several --pmc passes, one summary line per set)
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
SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_F32
SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VALU SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_LDS
TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCR_TCP_STALL_CYCLES_sum
TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum
TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TA_BUSY_avr
TD_TD_BUSY_sum TD_TC_STALL_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TCP_GATE_EN1_sum
TCP_TA_TCP_STATE_READ_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_GATE_EN2_sum TCP_TD_TCP_STALL_CYCLES_sum
SQ_INSTS_SALU SQ_INSTS_VALU SQ_INSTS_SMEM SQ_WAIT_INST_LDS
SETS
cat $out
