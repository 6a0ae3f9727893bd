#!/bin/bash
# 16 clips + encode leg: where the encode goes (behind the sweep on a branch / serial / at the start), stream priorities
cd "$(dirname "$0")/../.."
O=gpurun_out/r05e2; mkdir -p $O; export TMPDIR=/tmp; R=$PWD
python -c "import torch; print(torch.cuda.Stream.priority_range())" > $O/prio_range.txt 2>&1
for prec in f32 f16x3; do
  for cfg in "sweep_end 0 0" "serial 0 0" "start 0 0" "sweep_end -1 0" "sweep_end 0 -1" "start -1 0"; do
    set -- $cfg
    QPG_ENCODE_AT=$1 QPG_GRAPH_PRIO=$2 QPG_ENC_PRIO=$3 QPG_LOOP_CLIPS=16 QPG_LOOP_F16=1 QPG_LOOP_ENC=96 QPG_LOOP_ENC_PREC=$prec python tools/step_loop.py 40 graph 2>&1 | tail -1 | sed "s/^/prec=$prec encode_at=$1 graph_prio=$2 enc_prio=$3 /" >> $O/loops.log
  done
done
QPG_LOOP_CLIPS=16 QPG_LOOP_F16=1 python tools/step_loop.py 40 graph 2>&1 | tail -1 | sed "s/^/no encode /" >> $O/loops.log
cat $O/prio_range.txt $O/loops.log
