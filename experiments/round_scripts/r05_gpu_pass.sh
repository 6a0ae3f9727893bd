#!/bin/bash
# Round-5 evidence pass (everything under gpurun_out/r05e/): GPU suite, smoke, default bench line, rocprofv3 kernel stats of
# the driver's command, graph step timeline, PMC traffic of both split-f16 sweeps (two-plane f32 track, one-plane f16 track),
# BASELINE configs[4] (16 clips, f16 features, encode leg inside the captured step), configs[2], configs[3] on one GPU,
# the row-shard path on one rank with the library's own collectives.  usage: experiments/round_scripts/r05_gpu_pass.sh [quick]
set -u
O=gpurun_out/r05e; mkdir -p $O; export TMPDIR=/tmp; R=$PWD
QUICK=${1:-}
if [ "$QUICK" != "quick" ]; then
  timeout 3000 python -m pytest tests -m gpu -q > $O/pytest.log 2>&1; echo "pytest rc=$?" > $O/rc.txt
  timeout 600 python -c "import __graft_entry__ as g; g.build(); g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1; echo "smoke rc=$?" >> $O/rc.txt
else
  : > $O/rc.txt
fi
timeout 900 python bench.py --steps 20 --warmup 5 > $O/bench_20.json 2> $O/bench_20.err; echo "bench20 rc=$?" >> $O/rc.txt
timeout 900 python bench.py --no-cpu-baseline --no-e2e > $O/bench_200.json 2> $O/bench_200.err; echo "bench200 rc=$?" >> $O/rc.txt
( cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof -o bench -- python $R/bench.py --steps 20 --warmup 5 > $R/$O/bench_profiled.json 2> $R/$O/prof.err ); echo "prof rc=$?" >> $O/rc.txt
python tools/make_profile_summary.py $O/prof $O/bench_n1 "python bench.py --steps 20 --warmup 5 (N=1) under rocprofv3 --kernel-trace --stats" > /dev/null 2>&1
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --output-format csv -d $R/$O/tlg -- python $R/tools/step_loop.py 30 graph > $R/$O/tlg.log 2>&1 )
python tools/step_timeline.py $O/tlg 30 > $O/step_timeline_graph.md 2>&1
# configs[4]: 16 clips, f16 features, 96 pose windows encoded inside the captured step
timeout 900 python bench.py --steps 30 --warmup 5 --clips 16 --feature-dtype f16 --encode-batch 96 --no-vqvae --no-cold --no-e2e --no-cpu-baseline > $O/bench_clips16_f16_enc96.json 2> $O/bench_clips16_f16_enc96.err; echo "c16 f16 enc96 rc=$?" >> $O/rc.txt
if [ "$QUICK" = "first" ]; then cat $O/rc.txt; tail -3 $O/pytest.log; head -c 1500 $O/bench_20.json; echo; head -c 1200 $O/bench_clips16_f16_enc96.json; echo; tail -20 $O/step_timeline_graph.md; find $O -name "*kernel_trace.csv" -delete; find $O -name "*agent_info.csv" -delete; exit 0; fi
timeout 900 python bench.py --steps 30 --warmup 5 --clips 16 --feature-dtype f16 --encode-batch 96 --encode-precision f16x3 --no-vqvae --no-cold --no-e2e --no-cpu-baseline > $O/bench_clips16_f16_enc96_f16x3.json 2> $O/bench_clips16_f16_enc96_f16x3.err; echo "c16 f16 enc96 f16x3 rc=$?" >> $O/rc.txt
timeout 900 python bench.py --steps 30 --warmup 5 --clips 16 --feature-dtype f16 --no-vqvae --no-cold --no-e2e --no-cpu-baseline > $O/bench_clips16_f16.json 2> $O/bench_clips16_f16.err; echo "c16 f16 rc=$?" >> $O/rc.txt
timeout 900 python bench.py --steps 30 --warmup 5 --clips 16 --no-vqvae --no-cold --no-e2e --no-cpu-baseline > $O/bench_clips16_f32.json 2> $O/bench_clips16_f32.err; echo "c16 f32 rc=$?" >> $O/rc.txt
timeout 900 python bench.py --steps 100 --warmup 10 --feature-dtype f16 --no-vqvae --no-cold --no-e2e --no-cpu-baseline > $O/bench_clip1_f16.json 2> $O/bench_clip1_f16.err; echo "c1 f16 rc=$?" >> $O/rc.txt
( cd /tmp && QPG_LOOP_CLIPS=16 QPG_LOOP_F16=1 QPG_LOOP_ENC=96 timeout 600 rocprofv3 --kernel-trace --output-format csv -d $R/$O/tl16 -- python $R/tools/step_loop.py 20 graph > $R/$O/tl16.log 2>&1 )
python tools/step_timeline.py $O/tl16 20 > $O/step_timeline_c16_f16_enc96_graph.md 2>&1
# PMC traffic of the sweeps: Q=48 two-plane (f32 track), Q=768 one-plane (f16 track)
for c in FETCH_SIZE WRITE_SIZE; do
  ( cd /tmp && timeout 600 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $R/$O/pmc_$c -o a -- python $R/tools/bench_audio_hl.py 2048 48 > $R/$O/pmc_$c.log 2>&1 ); echo "pmc $c rc=$?" >> $O/rc.txt
  python tools/pmc_summary.py $O/pmc_$c audio > $O/pmc_$c.txt 2>&1
  ( cd /tmp && QPG_AUDIO_F16=1 timeout 600 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $R/$O/pmc16_$c -o a -- python $R/tools/bench_audio_hl.py 2048 768 > $R/$O/pmc16_$c.log 2>&1 ); echo "pmc16 $c rc=$?" >> $O/rc.txt
  python tools/pmc_summary.py $O/pmc16_$c audio > $O/pmc16_$c.txt 2>&1
done
python tools/pmc_traffic.py $O audio_cosine_hl2_kernel "N_db=2048 Q=48" $O/pmc_traffic.json > $O/pmc_traffic.txt 2>&1
timeout 900 python bench.py --scaling strong --no-cpu-baseline --no-vqvae > $O/bench_strong.json 2> $O/bench_strong.err; echo "strong rc=$?" >> $O/rc.txt
timeout 900 python bench.py --workload cfg3 > $O/bench_cfg3.json 2> $O/bench_cfg3.err; echo "cfg3 rc=$?" >> $O/rc.txt
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof_cfg3 -o cfg3 -- python $R/bench.py --workload cfg3 > $R/$O/bench_cfg3_profiled.json 2> $R/$O/prof_cfg3.err )
python tools/make_profile_summary.py $O/prof_cfg3 $O/cfg3 "python bench.py --workload cfg3 under rocprofv3 --kernel-trace --stats" > /dev/null 2>&1
python tools/bench_cfg3_parts2.py > $O/cfg3_parts.log 2>&1
bash tools/pmc_cfg3.sh > $O/pmc_cfg3.log 2>&1; cp gpurun_out/cfg3pmc/pmc_traffic_cfg3.json gpurun_out/cfg3pmc/FETCH_SIZE.txt gpurun_out/cfg3pmc/WRITE_SIZE.txt $O/ 2>/dev/null
python tools/bench_decode.py > $O/decode.log 2>&1
python tools/bench_vqvae.py > $O/vqvae.log 2>&1
timeout 300 python tools/bench_train.py 256 > $O/bench_train.log 2>&1
timeout 900 python bench.py --data speechlike --no-cpu-baseline --no-vqvae --no-e2e > $O/bench_speechlike.json 2> $O/bench_speechlike.err; echo "speechlike rc=$?" >> $O/rc.txt
for sc in weak strong; do
  for lc in 1 0; do
    QPG_LIB_COLLECTIVES=$lc QPG_BENCH_FORCE_SHARDED=1 MASTER_PORT=2955$lc timeout 600 python bench.py --gpus 1 --scaling $sc --n-db 2048 --steps 100 --warmup 10 --sharded-mixed-min-gflop 0 --check --no-vqvae --no-cold --no-e2e --no-cpu-baseline > $O/bench_forced_sharded_${sc}_lib$lc.json 2> $O/bench_forced_sharded_${sc}_lib$lc.err; echo "forced $sc lib$lc rc=$?" >> $O/rc.txt
  done
done
if [ -x experiments/rccl_graph/repro ]; then
  { for m in 0 1; do LD_LIBRARY_PATH=/opt/rocm/lib timeout 120 experiments/rccl_graph/repro $m 20; echo "rc=$?"; done; } > $O/rccl_repro.log 2>&1
fi
python tools/bench_db_cache.py > $O/db_cache.log 2>&1
python tools/bench_conv16.py > $O/conv16.log 2>&1
find $O -name "*.csv" -size +4M -delete; find $O -name "*kernel_trace.csv" -delete; find $O -name "*counter_collection.csv" -delete; find $O -name "*agent_info.csv" -delete
cat $O/rc.txt; [ -f $O/pytest.log ] && tail -3 $O/pytest.log
python - <<'P'
import json,glob
for f in sorted(glob.glob("gpurun_out/r05e/bench_*.json")):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        r=d.get("roofline",{})
        print(f.split("/")[-1], d["ms_per_step"], d.get("step_mode"), d.get("check"), "roof", r.get("bound"), r.get("frac"), r.get("kernel_ms"), "eager", d.get("eager",{}).get("ms_per_step"), "e2e", (d.get("e2e_cli") or {}).get("seconds"))
    except Exception as e:
        print(f, "ERR", e)
P
tail -18 $O/step_timeline_graph.md; cat $O/pmc_traffic.txt $O/pmc_FETCH_SIZE.txt $O/pmc16_FETCH_SIZE.txt 2>/dev/null | tail -30
