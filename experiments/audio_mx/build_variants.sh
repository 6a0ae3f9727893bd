#!/bin/bash
# Variant builds of the mixed-precision audio sweep for tools/bench_audio.py.  The padding / rotation / timing / lane-stride
# macros only exist in qpg_audio_instrumented.hip.txt (copy it over csrc/qpg_audio.hip to use them):
#   QPG_LIB_PATH=experiments/audio_mx/lib_<tag>.so python tools/bench_audio.py 2048 48 10 mx
set -e
cd "$(dirname "$0")/../.."
python -m qpgesture_amd.build > /dev/null
OBJS=$(ls qpgesture_amd/csrc/*.o | grep -v qpg_audio.o)
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt -fno-fast-math -Wno-unused-function"
for v in "$@"; do
  IFS=_ read occ ad mt mpl probe bd rot ks gs padf padq timing lstride org occ2 probe2 <<< "$v"
  /opt/rocm/bin/hipcc $FLAGS -DQPG_MX_OCC=$occ -DQPG_MX_AD=$ad -DQPG_MX_MT=$mt -DQPG_MX_MPL=${mpl:-2} -DQPG_MX_PROBE=${probe:-0} -DQPG_MX_BD=${bd:-2} -DQPG_MX_ROT=${rot:-1} -DQPG_MX_KS=${ks:-4} -DQPG_MX_GS=${gs:-1} -DQPG_MX_PADF=${padf:-0} -DQPG_MX_PADQ=${padq:-0} -DQPG_MX_TIMING=${timing:-0} -DQPG_MX_LSTRIDE=${lstride:-4} -DQPG_MX_ORG=${org:-2} -DQPG_MX2_OCC=${occ2:-2} -DQPG_MX2_PROBE=${probe2:-0} -c qpgesture_amd/csrc/qpg_audio.hip -o /tmp/qpg_audio_$v.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o experiments/audio_mx/lib_$v.so /tmp/qpg_audio_$v.o $OBJS
  echo built $v
done
