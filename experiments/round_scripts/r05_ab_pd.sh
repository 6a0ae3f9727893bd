cd "$(dirname "$0")/../.."
for r in 1 2; do
python tools/bench_cfg3_parts2.py 2>&1 | grep "gemm64h  " | sed "s/^/PD=2 /"
for v in 3 5; do QPG_LIB_PATH=experiments/gemm32/libqpg_pd$v.so python tools/bench_cfg3_parts2.py 2>&1 | grep "gemm64h  \|flag=" | sed "s/^/PD=$v /"; done
done
