#!/bin/bash
# A/B of the pre-launched replay behind the doorbell (round 6): bench.py's driver command with QPG_BENCH_DOORBELL=1 / 0,
# alternating, and graph timelines of tools/step_loop.py with / without it.  -> gpurun_out/r06_door
set -u
export TMPDIR=/tmp; R=$PWD; O=gpurun_out/r06_door; mkdir -p $O
python -m pytest tests/test_gpu_graph_pipeline.py -x -q 2>&1 | tail -3
for rep in 1 2 3; do for v in 1 0; do
  QPG_BENCH_DOORBELL=$v python bench.py --steps 20 --warmup 5 --no-sub-records --no-vqvae --no-cold --no-e2e --no-cpu-baseline > $O/line_${v}_$rep.json 2> $O/line_${v}_$rep.err
  python -c "
import json; d=json.loads(open('$O/line_${v}_$rep.json').read().strip().splitlines()[-1]); print('doorbell $v rep $rep: ms_per_step', d['ms_per_step'], 'eager', d['eager']['ms_per_step'], 'kernel', d['roofline']['kernel_ms'], d['graph_replay']['next_replay_prelaunched_behind_a_doorbell'], d['graph_replay']['other_seed_equals_eager'], d['mixed_precision']['codes_equal_f64_sweep'])"
done; done
for v in 1 0; do
  ( cd /tmp && QPG_LOOP_DOORBELL=$v timeout 300 rocprofv3 --kernel-trace --output-format csv -d $R/$O/tl_$v -- python $R/tools/step_loop.py 40 graph > $R/$O/tl_$v.log 2>&1 )
  tail -1 $O/tl_$v.log
done
find $O -name "*.csv" -delete
