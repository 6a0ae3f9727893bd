#!/bin/bash
# Round 6: the captured step with the MAIN (audio) branch on a high-priority stream, the text side's branch on a normal one
# (QPG_GRAPH_PRIO=-1): does the row streaming stop paying for the text GEMM beside it?  bench.py lines, alternating.
set -u
O=gpurun_out/r06_prio; mkdir -p $O
for rep in 1 2 3; do for v in 0 -1; do
  QPG_GRAPH_PRIO=$v python bench.py --steps 50 --warmup 5 --no-sub-records --no-vqvae --no-cold --no-e2e --no-cpu-baseline --no-f64-line > $O/l.json 2> $O/l.err
  python -c "
import json; d=json.loads(open('$O/l.json').read().strip().splitlines()[-1]); print('prio $v rep $rep: ms_per_step', d['ms_per_step'], 'kernel', d['roofline']['kernel_ms'], d['mixed_precision']['codes_equal_f64_sweep'])"
done; done
