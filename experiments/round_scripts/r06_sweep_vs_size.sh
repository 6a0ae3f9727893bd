#!/bin/bash
# the sweep at N_db = 2048 / 4096 / 8192, warm and cold; the read-stream ceiling at the three image
# sizes; FETCH_SIZE / WRITE_SIZE of the sweep at N = 8192 (separate PMC passes).  -> gpurun_out/r06_size
set -u
export TMPDIR=/tmp; R=$PWD; O=gpurun_out/r06_size; mkdir -p $O
python tools/sweep_vs_size.py > $O/sweep_vs_size.md 2> $O/sweep_vs_size.err; cat $O/sweep_vs_size.md
( cd experiments/hbm_read && [ -x read_bw ] || /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o read_bw read_bw.hip )
for mb in 691 1381 2762; do experiments/hbm_read/read_bw $mb | grep -i "own\|nontemporal" ; done > $O/read_bw.txt 2>&1; cat $O/read_bw.txt
for c in FETCH_SIZE WRITE_SIZE; do
  ( cd /tmp && timeout 600 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $R/$O/pmc_$c -o a -- python $R/tools/bench_audio_hl.py 8192 48 > $R/$O/pmc_$c.log 2>&1 )
done
python tools/pmc_traffic.py $O "audio_cosine_hl2_kernel<2" "N_db=8192 Q=48" $O/pmc_traffic_8192.json audio_cosine_hl2_kernel > $O/pmc_8192.txt 2>&1; cat $O/pmc_8192.txt
find $O -name "*.csv" -delete
