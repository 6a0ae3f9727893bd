#!/bin/bash
cd "$(dirname "$0")/../.."
O=gpurun_out/r05; mkdir -p $O; export TMPDIR=/tmp; R=$PWD
timeout 1200 python -m pytest tests/test_gpu_db_cache.py tests/test_gpu_matching.py tests/test_gpu_fullsize.py -x -q -m gpu -k "cache or cli or captured" > $O/pass3_tests.log 2>&1
echo "tests rc=$?" >> $O/pass3_tests.log
python tools/bench_db_cache.py > $O/pass3_db_cache.log 2>&1
for at in sweep_end start; do
  QPG_ENCODE_AT=$at QPG_LOOP_CLIPS=16 QPG_LOOP_F16=1 QPG_LOOP_ENC=96 python tools/step_loop.py 40 graph 2>&1 | tail -1 | sed "s/^/encode_at=$at clips=16 f16=1 enc=96 /" >> $O/pass3_loops.log
done
( cd /tmp && QPG_LOOP_CLIPS=16 QPG_LOOP_F16=1 QPG_LOOP_ENC=96 timeout 600 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $R/$O/tl16c -- python $R/tools/step_loop.py 20 graph > $R/$O/tl16c.log 2>&1 )
python tools/step_timeline.py $O/tl16c 20 > $O/pass3_timeline_c16_f16_enc96_graph.md 2>&1
find $O -name "*.csv" -delete
python bench.py --steps 50 --warmup 5 --no-vqvae --no-cold --no-cpu-baseline --no-f64-line > $O/pass3_bench_e2e.json 2> $O/pass3_bench_e2e.err
tail -3 $O/pass3_tests.log; cat $O/pass3_db_cache.log $O/pass3_loops.log; python -c "
import json; d=json.loads(open('$O/pass3_bench_e2e.json').read().strip().splitlines()[-1]); print(d['ms_per_step'], json.dumps(d.get('e2e_cli'), indent=1))"
