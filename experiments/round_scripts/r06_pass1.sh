#!/bin/bash
# Round 6, first evidence pass: the bench line (pipelined = GraphPipeline, sub_records), the same under rocprofv3 --stats,
# the one-clip graph timeline, the GraphPipeline timeline (overlap), the power-manager watch.  -> gpurun_out/r06_p1
set -u
export TMPDIR=/tmp; R=$PWD; O=gpurun_out/r06_p1; mkdir -p $O
python bench.py --steps 20 --warmup 5 > $O/bench_n1_line_steps20.json 2> $O/bench_n1_steps20.err
python bench.py > $O/bench_n1_line.json 2> $O/bench_n1.err
( cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof -o a -- python $R/bench.py --no-sub-records --no-cpu-baseline --no-cold --no-e2e > $R/$O/bench_n1_line_profiled.json 2> $R/$O/prof.err )
python tools/make_profile_summary.py $O/prof > $O/bench_n1_summary.md 2>&1 || true
cp $(find $O/prof -name "*kernel_stats.csv" | head -1) $O/bench_n1_kernel_stats.csv 2>/dev/null
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --output-format csv -d $R/$O/tl -- python $R/tools/step_loop.py 40 graph > $R/$O/tl.log 2>&1 )
python tools/step_timeline.py $O/tl 40 > $O/step_timeline_graph.md 2>&1
for c in 4:2 8:2; do
  t=${c/:/x}
  ( cd /tmp && timeout 600 rocprofv3 --kernel-trace --output-format csv -d $R/$O/pl_$t -- python $R/tools/bench_graph_pipeline.py $c > $R/$O/pl_$t.log 2>&1 )
  python tools/pipeline_timeline.py $O/pl_$t 1700 > $O/pipeline_timeline_$t.md 2>&1
done
python tools/clock_watch.py > $O/clock_watch.txt 2>&1
find $O -name "*.csv" ! -name "bench_n1_kernel_stats.csv" -delete
head -c 1500 $O/bench_n1_line_steps20.json; echo; python - <<'PY'
import json
d=json.loads(open("gpurun_out/r06_p1/bench_n1_line_steps20.json").read().strip().splitlines()[-1])
print("steps20:", d["ms_per_step"], d["value"], d["roofline"]["frac"], d["roofline"]["kernel_ms"])
print("pipelined:", json.dumps(d.get("pipelined"))[:1500])
d=json.loads(open("gpurun_out/r06_p1/bench_n1_line.json").read().strip().splitlines()[-1])
print("default:", d["ms_per_step"], d["value"], d["roofline"]["frac"], d["roofline"]["kernel_ms"], {k:v.get("ms_per_step") for k,v in d.get("sub_records",{}).items()})
PY
head -12 $O/pipeline_timeline_4x2.md; cat $O/clock_watch.txt | head -40; grep -i "span\|period" $O/step_timeline_graph.md
