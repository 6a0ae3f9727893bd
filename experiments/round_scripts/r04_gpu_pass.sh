#!/bin/bash
# Round-4 evidence pass (everything under gpurun_out/r04/): default bench line (20/5 and 200 steps), rocprofv3 kernel stats
# of the driver's command, eager + graph step timelines, PMC traffic of the sweep, the other workloads' lines, the
# row-shard path on one rank over RCCL.
set -u
O=gpurun_out/r04; mkdir -p $O; export TMPDIR=/tmp; R=$PWD
timeout 1500 python -m pytest tests -m gpu -q > $O/pytest.log 2>&1; echo "pytest rc=$?" > $O/rc.txt
timeout 600 python -c "import __graft_entry__ as g; g.build(); g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1; echo "smoke rc=$?" >> $O/rc.txt
timeout 900 python bench.py --steps 20 --warmup 5 > $O/bench_20.json 2> $O/bench_20.err; echo "bench20 rc=$?" >> $O/rc.txt
timeout 900 python bench.py --no-cpu-baseline --no-e2e > $O/bench_200.json 2> $O/bench_200.err; echo "bench200 rc=$?" >> $O/rc.txt
( cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof -o bench -- python $R/bench.py --steps 20 --warmup 5 > $R/$O/bench_profiled.json 2> $R/$O/prof.err ); echo "prof rc=$?" >> $O/rc.txt
python tools/make_profile_summary.py $O/prof $O/bench_n1 "python bench.py --steps 20 --warmup 5 (N=1) under rocprofv3 --kernel-trace --stats" > /dev/null 2>&1
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --output-format csv -d $R/$O/tl -- python $R/tools/step_loop.py 30 > $R/$O/tl.log 2>&1 )
python tools/step_timeline.py $O/tl 30 > $O/step_timeline_eager.md 2>&1
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --output-format csv -d $R/$O/tlg -- python $R/tools/step_loop.py 30 graph > $R/$O/tlg.log 2>&1 )
python tools/step_timeline.py $O/tlg 30 > $O/step_timeline_graph.md 2>&1
for c in FETCH_SIZE WRITE_SIZE; do
  ( cd /tmp && timeout 600 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $R/$O/pmc_$c -o a -- python $R/tools/bench_audio_hl.py > $R/$O/pmc_$c.log 2>&1 ); echo "pmc $c rc=$?" >> $O/rc.txt
  python tools/pmc_summary.py $O/pmc_$c audio > $O/pmc_$c.txt 2>&1
done
python tools/pmc_traffic.py $O audio_cosine_hl2_kernel "N_db=2048 Q=48" $O/pmc_traffic.json > $O/pmc_traffic.txt 2>&1
for s in SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_LDS SQ_INSTS_VALU SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE; do
  ( cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc $s --output-format csv -d $R/$O/sq_$s -o a -- python $R/tools/bench_audio_hl.py > /dev/null 2>&1 )
  python tools/pmc_summary.py $O/sq_$s audio_cosine_hl2 >> $O/pmc_issue_mix.txt 2>&1
done
timeout 900 python bench.py --clips 16 --no-cpu-baseline --no-vqvae > $O/bench_clips16.json 2> $O/bench_clips16.err; echo "clips16 rc=$?" >> $O/rc.txt
timeout 900 python bench.py --scaling strong --no-cpu-baseline --no-vqvae > $O/bench_strong.json 2> $O/bench_strong.err; echo "strong rc=$?" >> $O/rc.txt
timeout 900 python bench.py --workload cfg3 > $O/bench_cfg3.json 2> $O/bench_cfg3.err; echo "cfg3 rc=$?" >> $O/rc.txt
timeout 900 python bench.py --data speechlike --no-cpu-baseline --no-vqvae --no-e2e > $O/bench_speechlike.json 2> $O/bench_speechlike.err; echo "speechlike rc=$?" >> $O/rc.txt
for sc in weak strong; do
  QPG_BENCH_FORCE_SHARDED=1 timeout 600 python bench.py --steps 50 --warmup 5 --scaling $sc --n-db 2048 --no-cpu-baseline --no-vqvae --no-cold --check > $O/bench_forced_sharded_$sc.json 2> $O/bench_forced_sharded_$sc.err; echo "forced $sc rc=$?" >> $O/rc.txt
  QPG_BENCH_SHARDED_EAGER=1 QPG_BENCH_FORCE_SHARDED=1 timeout 600 python bench.py --steps 50 --warmup 5 --scaling $sc --n-db 2048 --no-cpu-baseline --no-vqvae --no-cold --check > $O/bench_forced_sharded_${sc}_eager.json 2> $O/bench_forced_sharded_${sc}_eager.err; echo "forced eager $sc rc=$?" >> $O/rc.txt
done
QPG_FORCE_SHARDED=1 QPG_EXPERIMENTAL_SHARDED_GRAPH=1 timeout 200 python tools/step_loop.py 200 graph > $O/forced_sharded_graph_loop.txt 2>&1
QPG_FORCE_SHARDED=1 timeout 200 python tools/step_loop.py 200 graph > $O/forced_sharded_segments_loop.txt 2>&1
QPG_FORCE_SHARDED=1 timeout 200 python tools/step_loop.py 200 > $O/forced_sharded_eager_loop.txt 2>&1
( cd /tmp && QPG_FORCE_SHARDED=1 timeout 600 rocprofv3 --kernel-trace --output-format csv -d $R/$O/tls -- python $R/tools/step_loop.py 30 graph > $R/$O/tls.log 2>&1 )
python tools/step_timeline.py $O/tls 30 > $O/step_timeline_forced_sharded.md 2>&1
( cd /tmp && QPG_FORCE_SHARDED=1 timeout 600 rocprofv3 --kernel-trace --output-format csv -d $R/$O/tlse -- python $R/tools/step_loop.py 30 > $R/$O/tlse.log 2>&1 )
python tools/step_timeline.py $O/tlse 30 > $O/step_timeline_forced_sharded_eager.md 2>&1
find $O -name "*.csv" -size +4M -delete; find $O -name "*kernel_trace.csv" -delete; find $O -name "*counter_collection.csv" -delete; find $O -name "*agent_info.csv" -delete
tail -2 $O/pytest.log; grep "ms/step" $O/forced_sharded_*_loop.txt; cat $O/rc.txt; head -c 400 $O/bench_20.json; echo; tail -18 $O/step_timeline_graph.md; cat $O/pmc_traffic.txt; cat $O/pmc_issue_mix.txt
