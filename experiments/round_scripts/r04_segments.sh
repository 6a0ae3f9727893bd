#!/bin/bash
# Row shards as hipGraph segments + eager collectives: tests, one-rank RCCL lines, loops, timeline.  Every command under timeout.
R=$(pwd); O=gpurun_out/r04seg; mkdir -p $O; rm -f $O/rc.txt
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_gpu_bench_sharded.py tests/test_gpu_mixed.py -x -q -m gpu > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/rc.txt
for sc in weak strong; do
  QPG_BENCH_FORCE_SHARDED=1 timeout 600 python bench.py --steps 50 --warmup 5 --scaling $sc --n-db 2048 --no-cpu-baseline --no-vqvae --no-cold --check > $O/bench_forced_sharded_$sc.json 2> $O/bench_forced_sharded_$sc.err; echo "forced $sc rc=$?" >> $O/rc.txt
  QPG_BENCH_SHARDED_EAGER=1 QPG_BENCH_FORCE_SHARDED=1 timeout 600 python bench.py --steps 50 --warmup 5 --scaling $sc --n-db 2048 --no-cpu-baseline --no-vqvae --no-cold --check > $O/bench_forced_sharded_${sc}_eager.json 2> $O/bench_forced_sharded_${sc}_eager.err; echo "forced eager $sc rc=$?" >> $O/rc.txt
done
QPG_FORCE_SHARDED=1 timeout 200 python tools/step_loop.py 200 graph > $O/forced_sharded_segments_loop.txt 2>&1
QPG_FORCE_SHARDED=1 timeout 200 python tools/step_loop.py 200 > $O/forced_sharded_eager_loop.txt 2>&1
( cd /tmp && QPG_FORCE_SHARDED=1 timeout 600 rocprofv3 --kernel-trace --output-format csv -d $R/$O/tls -- python $R/tools/step_loop.py 30 graph > $R/$O/tls.log 2>&1 )
python tools/step_timeline.py $O/tls 30 > $O/step_timeline_forced_sharded_segments.md 2>&1
find $O -name "*.csv" -size +4M -delete; find $O -name "*kernel_trace.csv" -delete; find $O -name "*agent_info.csv" -delete
tail -5 $O/pytest.log; cat $O/rc.txt; grep "ms/step" $O/forced_sharded_*_loop.txt; for f in $O/bench_forced_*.json; do echo $f; head -c 330 $f; echo; done; cat $O/step_timeline_forced_sharded_segments.md
