#!/bin/bash
# the ablation builds of experiments/round_scripts/r05_probe_bits.sh on the ONE-plane sweep at Q = 768 (16 clips: 16 chunk passes per launch)
cd "$(dirname "$0")/../.."
O=gpurun_out/r05p; mkdir -p $O
for r in 1 2; do for v in $(ls experiments/audio_hl/libqpg_p*.so); do echo "== $v"; QPG_LIB_PATH=$v timeout 300 python tools/bench_audio_hl.py 2048 768 2>&1 | grep "hl1" | sed 's/.*sweep min/min/'; done; done > $O/probe_bits_q768.log 2>&1
paste - - < $O/probe_bits_q768.log
