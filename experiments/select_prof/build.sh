#!/bin/bash
# Section timing of the mixed-precision audio select: libqpg_hip.so with -DQPG_SELECT_PROF (qpg_select.hip stamps the
# wall clock at section boundaries in block 0).  usage: experiments/select_prof/build.sh; then
#   QPG_LIB_PATH=experiments/select_prof/libqpg_prof.so python experiments/select_prof/run.py
set -e
cd "$(dirname "$0")/../.."
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt -fno-fast-math -Wno-unused-function"
OBJS=$(ls qpgesture_amd/csrc/*.o | grep -v qpg_select.o)
/opt/rocm/bin/hipcc $FLAGS -DQPG_SELECT_PROF -DQPG_DEBUG_HOOKS -c qpgesture_amd/csrc/qpg_select.hip -o /tmp/qpg_select_prof.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o experiments/select_prof/libqpg_prof.so $OBJS /tmp/qpg_select_prof.o
echo built
