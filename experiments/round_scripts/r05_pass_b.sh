#!/bin/bash
cd "$(dirname "$0")/../.."
O=gpurun_out/r05b; mkdir -p $O; export TMPDIR=/tmp; R=$PWD
for cfg in "1 96 f32" "1 96 f16x3" "1 0 f32" "0 0 f32"; do
  set -- $cfg
  tag=f16_$1_enc$2_$3
  ( cd /tmp && QPG_LOOP_CLIPS=16 QPG_LOOP_F16=$1 QPG_LOOP_ENC=$2 QPG_LOOP_ENC_PREC=$3 timeout 600 rocprofv3 --kernel-trace --output-format csv -d $R/$O/tl_$tag -- python $R/tools/step_loop.py 20 graph > $R/$O/tl_$tag.log 2>&1 )
  python tools/step_timeline.py $O/tl_$tag 20 > $O/timeline_c16_$tag.md 2>&1
done
python tools/bench_conv16.py > $O/conv16.log 2>&1
python tools/bench_decode.py > $O/decode.log 2>&1
python tools/bench_vqvae.py > $O/vqvae.log 2>&1
find $O -name "*.csv" -delete
for f in $O/timeline_*.md; do echo "== $f"; tail -45 $f; done; cat $O/conv16.log | tail -20; tail -5 $O/decode.log; tail -8 $O/vqvae.log
