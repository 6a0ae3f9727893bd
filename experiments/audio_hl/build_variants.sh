#!/bin/bash
# Ablation builds of the split-operand f16 sweep: one library per QPG_HL_PROBE value (see csrc/qpg_audio_hl.hip).
# usage: experiments/audio_hl/build_variants.sh "0 1 2 4 8 16 31"   -> experiments/audio_hl/libqpg_hl_<bits>.so
set -e
cd "$(dirname "$0")/../.."
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt -fno-fast-math -Wno-unused-function"
OBJS=$(ls qpgesture_amd/csrc/*.o | grep -v qpg_audio_hl.o)
for b in ${1:-0 1 2 4 8 16}; do
  /opt/rocm/bin/hipcc $FLAGS -DQPG_HL_PROBE=$b ${EXTRA:-} -c qpgesture_amd/csrc/qpg_audio_hl.hip -o /tmp/qpg_audio_hl_$b.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o experiments/audio_hl/libqpg_hl_$b.so $OBJS /tmp/qpg_audio_hl_$b.o
  echo built $b
done
