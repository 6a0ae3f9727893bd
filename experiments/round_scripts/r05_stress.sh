#!/bin/bash
# Randomised parity stress of round 5's code (tools/stress_*.py), every run under timeout; tails into gpurun_out/r05stress/
O=gpurun_out/r05stress; mkdir -p $O; rm -f $O/summary.txt
run() { n=$1; shift; timeout 1200 "$@" > $O/$n.log 2>&1; echo "$n rc=$? : $(tail -1 $O/$n.log)" >> $O/summary.txt; }
run parity python tools/stress_parity.py 40
run mixed python tools/stress_mixed.py 40
run sharded_mixed python tools/stress_sharded_mixed.py 30
run text python tools/stress_text.py 45
run cut python tools/stress_cut.py 40
run vqvae python tools/stress_vqvae.py 16
cat $O/summary.txt; grep -c "by code" $O/text.log
