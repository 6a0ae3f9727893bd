#!/bin/bash
# bottom-up ablations of the split-f16 convolution (experiments/conv_probe/var/libqpg_c<bits>.so, -DC16_PROBE=<bits>: 1 no MFMAs,
# 2 no split + LDS stores, 4 no global requests, 8 no fragment reads, 16 no slice barrier; sgb0: hipcc's own issue order)
cd "$(dirname "$0")/../.."
O=gpurun_out/r05c; mkdir -p $O
for r in 1 2; do for v in $(ls experiments/conv_probe/var/libqpg_c*.so); do echo "== $v"; QPG_LIB_PATH=$v timeout 300 python tools/bench_conv16.py 2>&1 | grep "k3 512" | sed 's/.*conv16/conv16/'; done; done > $O/probe_conv16.log 2>&1
paste - - < $O/probe_conv16.log
