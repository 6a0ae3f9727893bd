#!/bin/bash
# issue-side counters of the cfg-3 step's two kernels (one counter per pass): gpurun_out/cfg3issue/summary.txt
set -u
export TMPDIR=/tmp; R=$PWD; O=gpurun_out/cfg3issue; mkdir -p $O; : > $O/summary.txt
for c in SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VALU GRBM_GUI_ACTIVE TCC_HIT_sum TCC_MISS_sum; do
  ( cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $R/$O/p_$c -o a -- python $R/tools/bench_cfg3_parts2.py > /dev/null 2>&1 )
  python tools/pmc_summary.py $O/p_$c gemm64h >> $O/summary.txt 2>&1
  python tools/pmc_summary.py $O/p_$c bycode >> $O/summary.txt 2>&1
done
find $O -name "*.csv" -delete
cat $O/summary.txt
