#!/bin/bash
# Round 6: does the fork edge behind the clip pack (6.4 us between the pack and the sweep) go away when the text side waits
# for the END of the sweep (text_after_sweep) instead of being enqueued unordered behind the pack?  Graph timelines.
set -u
export TMPDIR=/tmp; R=$PWD; O=gpurun_out/r06_after; mkdir -p $O
run() { tag=$1; shift
  ( cd /tmp && env "$@" timeout 300 rocprofv3 --kernel-trace --output-format csv -d $R/$O/tl_$tag -- python $R/tools/step_loop.py 60 graph > $R/$O/tl_$tag.log 2>&1 )
  python tools/step_timeline.py $O/tl_$tag 60 > $O/timeline_$tag.md 2>&1; find $O/tl_$tag -name "*.csv" -delete
  echo "== $tag: $(tail -1 $O/tl_$tag.log)"; grep -i "pack\|audio_cosine\|gemm16\|gate_table\|span" $O/timeline_$tag.md | cut -c1-110
}
for rep in 1 2; do
  run default_$rep QPG_X=0
  run after_fused_$rep QPG_AUDIO_FIRST=0 QPG_TEXT_AFTER=1
  run after_split_$rep QPG_AUDIO_FIRST=0 QPG_TEXT_AFTER=1 QPG_FUSED_PACK=0
done
