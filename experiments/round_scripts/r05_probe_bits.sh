#!/bin/bash
# bottom-up ablations of the two-plane sweep (experiments/audio_hl/libqpg_p<bits>.so from tools/build_variant.sh qpg_audio_hl p<bits> "-DH2_PROBE=<bits>" experiments/audio_hl): 1 no HBM stream,
# 2 no f64 flush, 8 no stage barrier, 16 no query-fragment reads from LDS, 32 no query staging; two alternating rounds
cd "$(dirname "$0")/../.."
O=gpurun_out/r05p; mkdir -p $O
[ -n "${SKIP_CHAIN:-}" ] || ( cd experiments/mfma_peak && timeout 200 ./mfma_f16_chain ) > $O/mfma_f16_chain.txt 2>&1
for r in 1 2; do for v in $(ls experiments/audio_hl/libqpg_p*.so); do echo "== $v"; QPG_LIB_PATH=$v timeout 300 python tools/bench_audio_hl.py 2048 48 2>&1 | grep "hl sweep" | sed 's/|  *mx.*//'; done; done > $O/probe_bits.log 2>&1
cat $O/probe_bits.log
