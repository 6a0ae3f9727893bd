#!/bin/bash
# per-kernel timeline of the matching step replayed as ONE hipGraph (rocprofv3 kernel trace of tools/step_loop.py N graph)
set -u
O=gpurun_out/g; mkdir -p $O; export TMPDIR=/tmp; R=$PWD
python tools/bench_graph.py > $O/graph.txt 2>&1; cat $O/graph.txt
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --output-format csv -d $R/$O/tl -- python $R/tools/step_loop.py 30 graph > $R/$O/tl.log 2>&1 )
tail -1 $O/tl.log
python tools/step_timeline.py $O/tl 30 > $O/graph_timeline.md 2>&1
find $O -name "*.csv" -delete
cat $O/graph_timeline.md
