#!/bin/bash
# kernel timing of the sweep's ordering variants (experiments/audio_hl/libqpg_v*.so), two rounds each, alternating
cd "$(dirname "$0")/../.."
O=gpurun_out/r05v; mkdir -p $O
for r in 1 2; do for v in experiments/audio_hl/libqpg_v*.so; do echo "== $v"; QPG_LIB_PATH=$v python tools/bench_audio_hl.py 2048 48 2>&1 | grep "hl sweep" | sed 's/|  *mx.*//'; done; done > $O/variants.log 2>&1
cat $O/variants.log
