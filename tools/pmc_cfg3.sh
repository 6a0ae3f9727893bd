#!/bin/bash
# HBM traffic of EVERY kernel of a cfg-3 step (FETCH_SIZE / WRITE_SIZE, one counter per pass) -> gpurun_out/cfg3pmc/summary.txt
# and the step's total into profiles-style json (tools/pmc_cfg3.py)
set -u
export TMPDIR=/tmp; R=$PWD; O=gpurun_out/cfg3pmc; mkdir -p $O
for c in FETCH_SIZE WRITE_SIZE; do
  ( cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $R/$O/pmc_$c -o a -- python $R/bench.py --workload cfg3 --steps 10 > /dev/null 2>&1 )
  python tools/pmc_summary.py $O/pmc_$c > $O/$c.txt 2>&1
done
python tools/pmc_cfg3.py $O $O/pmc_traffic_cfg3.json
find $O -name "*.csv" -delete
cat $O/FETCH_SIZE.txt | grep -i "gemm\|sorted\|bycode\|perm32\|finish\|normalize\|pack_cols"; cat $O/pmc_traffic_cfg3.json
