#!/bin/bash
# round 5, first GPU pass: the one-plane f16 sweep (tests + kernel timing), the re-addressed two-plane sweep against round 4's
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out/r05
export TMPDIR=/tmp
{
  echo "== kernel timing, this tree"; python tools/bench_audio_hl.py 2048 48; python tools/bench_audio_hl.py 2048 768
  echo "== kernel timing, round 4's qpg_audio_hl.hip"; QPG_LIB_PATH=experiments/audio_hl/libqpg_r4.so python tools/bench_audio_hl.py 2048 48
  QPG_LIB_PATH=experiments/audio_hl/libqpg_r4.so python tools/bench_audio_hl.py 2048 768
  echo "== again, this tree"; python tools/bench_audio_hl.py 2048 48
} > gpurun_out/r05/pass1_kernels.log 2>&1
timeout 1500 python -m pytest tests/test_gpu_fullsize.py tests/test_gpu_audio_hl.py tests/test_gpu_matching.py -x -q -m gpu > gpurun_out/r05/pass1_tests.log 2>&1
echo "tests rc=$?" >> gpurun_out/r05/pass1_tests.log
python bench.py --steps 100 --warmup 10 --no-vqvae --no-cold --no-e2e --no-cpu-baseline > gpurun_out/r05/pass1_bench.json 2> gpurun_out/r05/pass1_bench.err
python bench.py --steps 30 --warmup 5 --clips 16 --feature-dtype f16 --no-vqvae --no-cold --no-e2e --no-cpu-baseline > gpurun_out/r05/pass1_bench_c16_f16.json 2> gpurun_out/r05/pass1_bench_c16_f16.err
python bench.py --steps 30 --warmup 5 --clips 16 --no-vqvae --no-cold --no-e2e --no-cpu-baseline > gpurun_out/r05/pass1_bench_c16_f32.json 2> gpurun_out/r05/pass1_bench_c16_f32.err
tail -5 gpurun_out/r05/pass1_kernels.log gpurun_out/r05/pass1_tests.log
