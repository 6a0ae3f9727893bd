#!/bin/bash
# A variant build of ONE kernel source into its own library (run with QPG_LIB_PATH=<that .so>): the product objects of every
# other source + this one compiled with extra defines.  What experiments/round_scripts/r05_probe_bits.sh, r05_ab_nt*.sh, r05_probe_gemm64.sh and
# r05_probe_conv16.sh run over.
#   tools/build_variant.sh qpg_audio_hl p122nt "-DH2_PROBE=122 -DH2_NT=1" experiments/audio_hl      -> .../libqpg_p122nt.so
#   tools/build_variant.sh qpg_audio_hl g1 "-DG64_PROBE=1" experiments/gemm32/var                   -> .../libqpg_pg1.so (name it pg1)
#   tools/build_variant.sh qpg_conv16 c1 "-DC16_PROBE=1" experiments/conv_probe/var                 -> .../libqpg_c1.so
set -e
cd "$(dirname "$0")/.."
SRC=$1; NAME=$2; DEFS=$3; OUT=${4:-experiments/variants}
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt -fno-fast-math -Wno-unused-function"
[ -f qpgesture_amd/csrc/qpg_core.o ] || python -m qpgesture_amd.build > /dev/null
OBJS=$(ls qpgesture_amd/csrc/*.o | grep -v "/$SRC.o")
mkdir -p $OUT
/opt/rocm/bin/hipcc $FLAGS $DEFS -c qpgesture_amd/csrc/$SRC.hip -o /tmp/${SRC}_$NAME.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $OUT/libqpg_$NAME.so $OBJS /tmp/${SRC}_$NAME.o
echo "built $OUT/libqpg_$NAME.so"
