#!/bin/bash
# sweep duration INSIDE the captured step (rocprofv3 kernel trace of graph replays), plain vs non-temporal fragment loads;
# MODE=MODE_AUD: the audio side alone (no text GEMM queued beside the sweep)
cd "$(dirname "$0")/../.."
export TMPDIR=/tmp; R=$PWD; O=gpurun_out/r05nt; mkdir -p $O; L=experiments/audio_hl
for m in ${MODES:-MODE_AUD_TXT MODE_AUD}; do for r in 1 2; do for v in ${VARIANTS:-nt0 nt1}; do
  rm -rf $O/tl_$v
  ( cd /tmp && QPG_LOOP_MODE=$m QPG_LIB_PATH=$R/$L/libqpg_p$v.so timeout 600 rocprofv3 --kernel-trace --output-format csv -d $R/$O/tl_$v -- python $R/tools/step_loop.py 60 graph > $R/$O/tl_$v.log 2>&1 )
  echo "== $m $v (round $r)"; python tools/step_timeline.py $O/tl_$v 60 2>&1 | grep "audio_cosine_hl2\|hl_gemm16\|mixed_stream\|percode_select_mixed\|GPU-side span"
done; done; done > $O/ab_nt_tl.log 2>&1
find $O -name "*.csv" -delete
cat $O/ab_nt_tl.log
