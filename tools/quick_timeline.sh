#!/bin/bash
# bench line + per-kernel step timeline (rocprofv3 kernel trace of tools/step_loop.py); env knobs of step_loop.py pass through
set -u
O=gpurun_out/q; mkdir -p $O; export TMPDIR=/tmp; R=$PWD
if [ "${SKIP_BENCH:-0}" != "1" ]; then
  timeout 600 python bench.py --no-f64-line > $O/bench.json 2> $O/bench.err
  python -c "
import json; d=json.load(open('$O/bench.json')); print({k:d[k] for k in ('value','ms_per_step')}, d['roofline'].get('kernel_ms'))"
fi
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $R/$O/tl -- python $R/tools/step_loop.py 30 > $R/$O/tl.log 2>&1 )
python tools/step_timeline.py $O/tl 30 > $O/step_timeline.md 2>&1
find $O -name "*.csv" -delete
cat $O/step_timeline.md
