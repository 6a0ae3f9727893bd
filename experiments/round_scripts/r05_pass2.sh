#!/bin/bash
# round 5, second GPU pass: captured 16-clip step (+ encode leg), tests of the touched paths, timelines
cd "$(dirname "$0")/../.."
O=gpurun_out/r05; mkdir -p $O; export TMPDIR=/tmp; R=$PWD
timeout 2400 python -m pytest tests/test_gpu_fullsize.py tests/test_gpu_audio_hl.py tests/test_gpu_matching.py tests/test_gpu_guard_overflow.py tests/test_gpu_mixed.py tests/test_gpu_bench_sharded.py -x -q -m gpu > $O/pass2_tests.log 2>&1
echo "tests rc=$?" >> $O/pass2_tests.log
for cfg in "16 1 0" "16 1 96" "16 0 0"; do
  set -- $cfg
  for mode in graph eager; do
    QPG_LOOP_CLIPS=$1 QPG_LOOP_F16=$2 QPG_LOOP_ENC=$3 python tools/step_loop.py 40 $mode 2>&1 | tail -1 | sed "s/^/clips=$1 f16=$2 enc=$3 /" >> $O/pass2_loops.log
  done
done
( cd /tmp && QPG_LOOP_CLIPS=16 QPG_LOOP_F16=1 QPG_LOOP_ENC=96 timeout 600 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $R/$O/tl16 -- python $R/tools/step_loop.py 20 graph > $R/$O/tl16.log 2>&1 )
python tools/step_timeline.py $O/tl16 20 > $O/pass2_timeline_c16_f16_enc96_graph.md 2>&1
( cd /tmp && QPG_LOOP_CLIPS=16 QPG_LOOP_F16=1 timeout 600 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $R/$O/tl16b -- python $R/tools/step_loop.py 20 graph > $R/$O/tl16b.log 2>&1 )
python tools/step_timeline.py $O/tl16b 20 > $O/pass2_timeline_c16_f16_graph.md 2>&1
find $O -name "*.csv" -delete
python bench.py --steps 30 --warmup 5 --clips 16 --feature-dtype f16 --encode-batch 96 --no-vqvae --no-cold --no-e2e --no-cpu-baseline > $O/pass2_bench_c16_f16_enc96.json 2> $O/pass2_bench_c16_f16_enc96.err
python bench.py --steps 30 --warmup 5 --clips 16 --feature-dtype f16 --no-vqvae --no-cold --no-e2e --no-cpu-baseline > $O/pass2_bench_c16_f16.json 2> $O/pass2_bench_c16_f16.err
python bench.py --steps 100 --warmup 10 --feature-dtype f16 --no-vqvae --no-cold --no-e2e --no-cpu-baseline > $O/pass2_bench_c1_f16.json 2> $O/pass2_bench_c1_f16.err
tail -4 $O/pass2_tests.log; cat $O/pass2_loops.log
