#!/bin/bash
cd "$(dirname "$0")/../.."
O=gpurun_out/r05d; mkdir -p $O; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_cfg3.py -x -q -m gpu -s -k "by_code or falls_back" > $O/tests.log 2>&1; echo "tests rc=$?" >> $O/tests.log
python tools/bench_cfg3_parts2.py > $O/parts.log 2>&1

grep -E "passed|failed|rc=|cfg-3|h-plane" $O/tests.log | tail; cat $O/parts.log | grep -v amdgpu
