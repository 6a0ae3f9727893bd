#!/bin/bash
# per-kernel statistics of the VQ-VAE training step at the reference's batch size (rocprofv3 --kernel-trace --stats of tools/bench_train.py)
set -u
O=gpurun_out/train; mkdir -p $O; export TMPDIR=/tmp; R=$PWD
python tools/bench_train.py 256 > $O/bench_train.log 2>&1; cat $O/bench_train.log
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof -- python $R/tools/bench_train.py 256 > $R/$O/prof.log 2>&1 )
f=$(find $O/prof -name "*kernel_stats.csv" | head -1)
cp "$f" $O/train_kernel_stats.csv
find $O/prof -name "*.csv" -delete
head -24 $O/train_kernel_stats.csv | cut -c1-160
