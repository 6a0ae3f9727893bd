#!/bin/bash
set -u
O=gpurun_out/g2; mkdir -p $O; export TMPDIR=/tmp; R=$PWD
for mode in "1 x" "0 0"; do
  set -- $mode
  export QPG_AUDIO_FIRST=$1; if [ "$2" = "x" ]; then unset QPG_TEXT_AFTER; else export QPG_TEXT_AFTER=$2; fi
  echo "== audio_first=$1 text_after=${2}"
  python tools/step_loop.py 200 graph 2>/dev/null | tail -1
  python tools/step_loop.py 200 2>/dev/null | tail -1
  ( cd /tmp && timeout 600 rocprofv3 --kernel-trace --output-format csv -d $R/$O/tl$1 -- python $R/tools/step_loop.py 30 graph > $R/$O/tl$1.log 2>&1 )
  python tools/step_timeline.py $O/tl$1 30 | tail -18
done
find $O -name "*.csv" -delete
