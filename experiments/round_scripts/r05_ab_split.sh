#!/bin/bash
# rank fusion per modality behind its own select (split_fuse, the default) against the one launch in the walk, alternating
cd "$(dirname "$0")/../.."
export TMPDIR=/tmp; R=$PWD; O=gpurun_out/r05s; mkdir -p $O
for v in 0 1 0 1 0 1; do QPG_SPLIT_FUSE=$v python tools/step_loop.py 300 graph 2>&1 | tail -1 | sed "s/^/split=$v clip1 /"; done
for v in 0 1 0 1; do QPG_SPLIT_FUSE=$v QPG_LOOP_CLIPS=16 QPG_LOOP_F16=1 python tools/step_loop.py 40 graph 2>&1 | tail -1 | sed "s/^/split=$v clips16 /"; done
for v in 0 1; do
  rm -rf $O/tl$v
  ( cd /tmp && QPG_SPLIT_FUSE=$v timeout 600 rocprofv3 --kernel-trace --output-format csv -d $R/$O/tl$v -- python $R/tools/step_loop.py 40 graph > $R/$O/tl$v.log 2>&1 )
  echo "== split=$v"; python tools/step_timeline.py $O/tl$v 40 2>&1 | tail -16
done
find $O -name "*.csv" -delete
