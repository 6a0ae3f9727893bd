#!/bin/bash
cd "$(dirname "$0")/../.."
O=gpurun_out/r05g; mkdir -p $O; export TMPDIR=/tmp; R=$PWD
timeout 1500 python -m pytest tests/test_gpu_fullsize.py tests/test_gpu_matching.py -x -q -m gpu > $O/tests.log 2>&1; echo "tests rc=$?" >> $O/tests.log
for e in 0 96; do QPG_LOOP_CLIPS=16 QPG_LOOP_F16=1 QPG_LOOP_ENC=$e python tools/step_loop.py 40 graph 2>&1 | tail -1 | sed "s/^/clips16 f16 enc=$e /"; done > $O/loops.log 2>&1
( cd /tmp && QPG_LOOP_CLIPS=16 QPG_LOOP_F16=1 timeout 600 rocprofv3 --kernel-trace --output-format csv -d $R/$O/tl16b -- python $R/tools/step_loop.py 20 graph > $R/$O/tl16b.log 2>&1 )
python tools/step_timeline.py $O/tl16b 20 > $O/step_timeline_c16_f16_graph.md 2>&1
find $O -name "*.csv" -delete
tail -3 $O/tests.log; cat $O/loops.log; tail -8 $O/step_timeline_c16_f16_graph.md
