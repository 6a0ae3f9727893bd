set -u
export TMPDIR=/tmp; R=$PWD; O=gpurun_out/r06_p2; mkdir -p $O
python -m pytest tests/test_gpu_cfg3.py tests/test_gpu_multi.py -x -q 2>&1 | tail -5 > $O/t.log; cat $O/t.log
python bench.py --workload cfg3 --steps 30 > $O/bench_cfg3.json 2> $O/cfg3.err; python -c "
import json; d=json.loads(open('$O/bench_cfg3.json').read().strip().splitlines()[-1]); print('cfg3', d['ms_per_step'], d['roofline']['kernel_ms'], d.get('tables_equal_exact_sweep'))"
python tools/bench_graph_pipeline.py > $O/pipe_sweep.txt 2>&1; cat $O/pipe_sweep.txt | tail -12
for c in 4:2:1 2:2:1; do
  t=${c//:/x}
  ( cd /tmp && timeout 600 rocprofv3 --kernel-trace --output-format csv -d $R/$O/pl_$t -- python $R/tools/bench_graph_pipeline.py $c > $R/$O/pl_$t.log 2>&1 )
  python tools/pipeline_timeline.py $O/pl_$t 2600 > $O/pipeline_timeline_$t.md 2>&1
  tail -1 $O/pl_$t.log
done
find $O -name "*.csv" -delete
head -70 $O/pipeline_timeline_4x2x1.md
