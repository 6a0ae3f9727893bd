#!/bin/bash
cd "$(dirname "$0")/../.."
O=gpurun_out/r05f; mkdir -p $O; export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_gpu_cfg3.py tests/test_gpu_fullsize.py tests/test_gpu_text_prefilter.py tests/test_gpu_db_cache.py -x -q -m gpu > $O/tests.log 2>&1; echo "tests rc=$?" >> $O/tests.log
for e in 0 96; do QPG_LOOP_CLIPS=16 QPG_LOOP_F16=1 QPG_LOOP_ENC=$e python tools/step_loop.py 40 graph 2>&1 | tail -1 | sed "s/^/clips16 f16 enc=$e /"; done > $O/loops.log 2>&1
QPG_LOOP_CLIPS=16 python tools/step_loop.py 40 graph 2>&1 | tail -1 | sed "s/^/clips16 f32 /" >> $O/loops.log
python tools/step_loop.py 100 graph 2>&1 | tail -1 | sed "s/^/clip1 /" >> $O/loops.log
tail -4 $O/tests.log; cat $O/loops.log
