#!/bin/bash
# Round-6 evidence pass (everything under gpurun_out/r06e/; copied into profiles/r06_* afterwards): GPU suite, smoke, the
# driver's bench line (+ the default one) with `pipelined` = GraphPipeline and `sub_records`, rocprofv3 kernel stats of the
# driver's command, the one-clip graph timeline, GraphPipeline sweep + timelines (lockstep / staggered), PMC traffic of the
# sweep (N = 2048 and 8192), sweep vs size (warm / cold), read-stream ceiling, the power-manager watch, cfg-3 line + kernel
# stats, the epilogue-atomics experiment, randomised parity stress.   usage: experiments/round_scripts/r06_gpu_pass.sh [quick]
set -u
O=gpurun_out/r06e; mkdir -p $O; export TMPDIR=/tmp; R=$PWD
QUICK=${1:-}
: > $O/rc.txt
if [ "$QUICK" != "quick" ]; then
  timeout 3000 python -m pytest tests -m gpu -q > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/rc.txt
  timeout 600 python -c "import __graft_entry__ as g; g.build(); g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1; echo "smoke rc=$?" >> $O/rc.txt
fi
timeout 900 python bench.py --steps 20 --warmup 5 > $O/bench_n1_line_steps20.json 2> $O/bench_20.err; echo "bench20 rc=$?" >> $O/rc.txt
timeout 900 python bench.py > $O/bench_n1_line.json 2> $O/bench_200.err; echo "bench200 rc=$?" >> $O/rc.txt
( cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof -o bench -- python $R/bench.py --steps 20 --warmup 5 --no-sub-records > $R/$O/bench_n1_line_profiled.json 2> $R/$O/prof.err ); echo "prof rc=$?" >> $O/rc.txt
python tools/make_profile_summary.py $O/prof $O/bench_n1 "python bench.py --steps 20 --warmup 5 --no-sub-records (N=1) under rocprofv3 --kernel-trace --stats" > /dev/null 2>&1
cp profiles/kernel_replay.json $O/kernel_replay.json
[ -f $O/bench_n1_kernel_stats.csv ] && python tools/kernel_replay.py $O/bench_n1_kernel_stats.csv "audio_cosine_hl2_kernel<2, 2, true>" "audio_cosine_hl2_kernel|N_db=2048 Q=48" "python bench.py --steps 20 --warmup 5 --no-sub-records" $O/kernel_replay.json > $O/kernel_replay.txt 2>&1
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --output-format csv -d $R/$O/tlg -- python $R/tools/step_loop.py 40 graph > $R/$O/tlg.log 2>&1 )
python tools/step_timeline.py $O/tlg 40 > $O/step_timeline_graph.md 2>&1
# GraphPipeline: the sweep over (clips per replay, lanes, stagger) and three timelines
timeout 900 python tools/bench_graph_pipeline.py > $O/pipe_sweep.txt 2>&1; echo "pipe sweep rc=$?" >> $O/rc.txt
for c in 4:2:0 4:2:1 8:2:0; do
  t=${c//:/x}
  ( cd /tmp && timeout 600 rocprofv3 --kernel-trace --output-format csv -d $R/$O/pl_$t -- python $R/tools/bench_graph_pipeline.py $c > $R/$O/pl_$t.log 2>&1 )
  python tools/pipeline_timeline.py $O/pl_$t 2600 > $O/pipeline_timeline_$t.md 2>&1
done
# PMC traffic of the sweep: N = 2048 (what bench.py reads roofline.traffic from) and N = 8192
cp profiles/pmc_traffic.json $O/pmc_traffic.json
for n in 2048 8192; do
  for c in FETCH_SIZE WRITE_SIZE; do
    ( cd /tmp && timeout 600 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $R/$O/pmc$n/pmc_$c -o a -- python $R/tools/bench_audio_hl.py $n 48 > $R/$O/pmc${n}_$c.log 2>&1 ); echo "pmc $n $c rc=$?" >> $O/rc.txt
  done
  python tools/pmc_traffic.py $O/pmc$n "audio_cosine_hl2_kernel<2" "N_db=$n Q=48" $O/pmc_traffic.json audio_cosine_hl2_kernel > $O/pmc_traffic_$n.txt 2>&1
done
python tools/sweep_vs_size.py > $O/sweep_vs_size.md 2> $O/sweep_vs_size.err; echo "sweep_vs_size rc=$?" >> $O/rc.txt
( cd experiments/hbm_read && { [ -x read_bw ] || /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o read_bw read_bw.hip; } )
for mb in 691 1381 2762; do experiments/hbm_read/read_bw $mb | grep -i "own\|nontemporal"; done > $O/read_bw.txt 2>&1
python tools/clock_watch.py > $O/clock_watch.txt 2>&1; echo "clock_watch rc=$?" >> $O/rc.txt
( cd experiments/epilogue_atomics && { [ -x epi_atomics ] || /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o epi_atomics epi_atomics.hip; } && ./epi_atomics ) > $O/epilogue_atomics.txt 2>&1
# cfg-3: the line, kernel stats, per-launch times
timeout 600 python bench.py --workload cfg3 > $O/bench_cfg3.json 2> $O/bench_cfg3.err; echo "cfg3 rc=$?" >> $O/rc.txt
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof_cfg3 -o cfg3 -- python $R/bench.py --workload cfg3 > $R/$O/bench_cfg3_profiled.json 2> $R/$O/prof_cfg3.err )
python tools/make_profile_summary.py $O/prof_cfg3 $O/cfg3 "python bench.py --workload cfg3 under rocprofv3 --kernel-trace --stats" > /dev/null 2>&1
python tools/bench_decode.py > $O/decode.log 2>&1
if [ "$QUICK" != "quick" ]; then
  run() { n=$1; shift; timeout 1200 "$@" > $O/stress_$n.log 2>&1; echo "stress $n rc=$? : $(tail -1 $O/stress_$n.log)" >> $O/stress_summary.txt; }
  rm -f $O/stress_summary.txt
  run parity python tools/stress_parity.py 30
  run mixed python tools/stress_mixed.py 30
  run text python tools/stress_text.py 30
  run cut python tools/stress_cut.py 30
  cat $O/stress_summary.txt
fi
find $O -name "*kernel_trace.csv" -delete; find $O -name "*counter_collection.csv" -delete; find $O -name "*agent_info.csv" -delete; find $O -name "*.csv" -size +4M -delete
find $O -type d -empty -delete
cat $O/rc.txt; [ -f $O/pytest.log ] && tail -2 $O/pytest.log
python - <<'P'
import json
for f in ("bench_n1_line_steps20", "bench_n1_line", "bench_n1_line_profiled", "bench_cfg3"):
    try:
        d = json.loads(open("gpurun_out/r06e/%s.json" % f).read().strip().splitlines()[-1])
        r = d.get("roofline", {})
        p = d.get("pipelined", {})
        print(f, d["ms_per_step"], d.get("value"), "roof", r.get("frac"), r.get("kernel_ms"), "pipelined", p.get("ms_per_step"),
              (p.get("deeper") or {}).get("ms_per_step"), "sub", {k: v.get("ms_per_step") for k, v in (d.get("sub_records") or {}).items()})
    except Exception as e:
        print(f, "ERR", e)
P
cat $O/kernel_replay.txt $O/pmc_traffic_2048.txt $O/pmc_traffic_8192.txt 2>/dev/null; cat $O/sweep_vs_size.md; tail -14 $O/pipe_sweep.txt; grep "post-sweep" $O/pipeline_timeline_*.md; head -30 $O/clock_watch.txt; grep -i "span" $O/step_timeline_graph.md
