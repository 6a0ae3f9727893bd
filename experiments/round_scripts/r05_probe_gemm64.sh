#!/bin/bash
# bottom-up ablations of the h-plane prefilter GEMM (experiments/gemm32/var/libqpg_pg<bits>.so, -DG64_PROBE=<bits>: 1 no
# epilogue, 2 no row stream, 4 no query staging, 8 no MFMAs, 16 no query-fragment reads), two alternating rounds
cd "$(dirname "$0")/../.."
O=gpurun_out/r05g; mkdir -p $O
for r in 1 2; do for v in $(ls experiments/gemm32/var/libqpg_pg*.so); do echo "== $v"; QPG_LIB_PATH=$v timeout 300 python tools/bench_gemm64h.py 2>&1 | tail -1; done; done > $O/probe_gemm64.log 2>&1
cat $O/probe_gemm64.log
