#!/bin/bash
# A/B of the streaming pass's merge (round 6): per-slice rows of minima written with plain stores (product) against round 5's
# atomicMax into one shared row (-DMIX_INV_ATOMIC=1: experiments/variants/libqpg_invatomic.so), graph timelines, alternating.
set -u
O=gpurun_out/r06_ab_stream; mkdir -p $O; export TMPDIR=/tmp; R=$PWD
for rep in 1 2; do
for v in product invatomic; do
  L=""; [ $v = invatomic ] && L=$R/experiments/variants/libqpg_invatomic.so
  ( cd /tmp && QPG_LIB_PATH=$L timeout 600 rocprofv3 --kernel-trace --output-format csv -d $R/$O/tl_${v}_$rep -- python $R/tools/step_loop.py 40 graph > $R/$O/tl_${v}_$rep.log 2>&1 )
  python tools/step_timeline.py $O/tl_${v}_$rep 40 > $O/timeline_${v}_$rep.md 2>&1
  find $O/tl_${v}_$rep -name "*.csv" -delete
  echo "== $v $rep"; tail -1 $O/tl_${v}_$rep.log; grep -i 'mixed_stream\|percode_select_mixed\|span\|period' $O/timeline_${v}_$rep.md | head -8
done; done
