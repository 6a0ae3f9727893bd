#!/bin/bash
# round 5, the pass behind the non-temporal fragment loads: PMC traffic of both sweeps first (bench.py reads it), the whole
# GPU suite, smoke, the driver's line + its rocprofv3 kernel stats, timelines, the 16-clip / f16 / cfg-3 lines
cd "$(dirname "$0")/../.."
O=gpurun_out/r05last; mkdir -p $O; export TMPDIR=/tmp; R=$PWD
: > $O/rc.txt
for c in FETCH_SIZE WRITE_SIZE; do
  ( cd /tmp && timeout 600 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $R/$O/pmc_$c -o a -- python $R/tools/bench_audio_hl.py 2048 48 > $R/$O/pmc_$c.log 2>&1 ); echo "pmc $c rc=$?" >> $O/rc.txt
  python tools/pmc_summary.py $O/pmc_$c audio > $O/pmc_$c.txt 2>&1
done
cp profiles/pmc_traffic.json $O/pmc_traffic.json
python tools/pmc_traffic.py $O "audio_cosine_hl2_kernel<2" "N_db=2048 Q=48" $O/pmc_traffic.json audio_cosine_hl2_kernel > $O/pmc_traffic.txt 2>&1
python tools/pmc_traffic.py $O "audio_cosine_hl2_kernel<1" "N_db=2048 Q=48" $O/pmc_traffic.json audio_cosine_hl1 >> $O/pmc_traffic.txt 2>&1
cp $O/pmc_traffic.json profiles/pmc_traffic.json
timeout 3000 python -m pytest tests -m gpu -q > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/rc.txt
timeout 600 python -c "import __graft_entry__ as g; g.build(); g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1; echo "smoke rc=$?" >> $O/rc.txt
( cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof -o bench -- python $R/bench.py --steps 20 --warmup 5 > $R/$O/bench_profiled.json 2> $R/$O/prof.err ); echo "prof rc=$?" >> $O/rc.txt
python tools/make_profile_summary.py $O/prof $O/bench_n1 "python bench.py --steps 20 --warmup 5 (N=1) under rocprofv3 --kernel-trace --stats" > /dev/null 2>&1
ST=$(find $O/prof -name "*kernel_stats.csv" | head -1)
[ -n "$ST" ] && python tools/kernel_replay.py $ST audio_cosine_hl2_kernel "audio_cosine_hl2_kernel|N_db=2048 Q=48" "python bench.py --steps 20 --warmup 5" profiles/kernel_replay.json > $O/kernel_replay.txt 2>&1
cp profiles/kernel_replay.json $O/kernel_replay.json
timeout 900 python bench.py --steps 20 --warmup 5 > $O/bench_20.json 2> $O/bench_20.err; echo "bench20 rc=$?" >> $O/rc.txt
timeout 900 python bench.py --no-cpu-baseline --no-e2e > $O/bench_200.json 2> $O/bench_200.err; echo "bench200 rc=$?" >> $O/rc.txt
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --output-format csv -d $R/$O/tlg -- python $R/tools/step_loop.py 30 graph > $R/$O/tlg.log 2>&1 )
python tools/step_timeline.py $O/tlg 30 > $O/step_timeline_graph.md 2>&1
timeout 900 python bench.py --steps 100 --warmup 10 --feature-dtype f16 --no-vqvae --no-cold --no-e2e --no-cpu-baseline > $O/bench_clip1_f16.json 2> $O/e0.err; echo "c1 f16 rc=$?" >> $O/rc.txt
timeout 900 python bench.py --steps 30 --warmup 5 --clips 16 --feature-dtype f16 --encode-batch 96 --no-vqvae --no-cold --no-e2e --no-cpu-baseline > $O/bench_clips16_f16_enc96.json 2> $O/e1.err; echo "c16 enc rc=$?" >> $O/rc.txt
timeout 900 python bench.py --steps 30 --warmup 5 --clips 16 --feature-dtype f16 --encode-batch 96 --encode-precision f16x3 --no-vqvae --no-cold --no-e2e --no-cpu-baseline > $O/bench_clips16_f16_enc96_f16x3.json 2> $O/e2.err; echo "c16 enc f16x3 rc=$?" >> $O/rc.txt
timeout 900 python bench.py --steps 30 --warmup 5 --clips 16 --feature-dtype f16 --no-vqvae --no-cold --no-e2e --no-cpu-baseline > $O/bench_clips16_f16.json 2> $O/e3.err; echo "c16 rc=$?" >> $O/rc.txt
timeout 900 python bench.py --steps 30 --warmup 5 --clips 16 --no-vqvae --no-cold --no-e2e --no-cpu-baseline > $O/bench_clips16_f32.json 2> $O/e4.err; echo "c16 f32 rc=$?" >> $O/rc.txt
timeout 900 python bench.py --workload cfg3 > $O/bench_cfg3.json 2> $O/e5.err; echo "cfg3 rc=$?" >> $O/rc.txt
timeout 900 python bench.py --data speechlike --no-cpu-baseline --no-vqvae --no-e2e > $O/bench_speechlike.json 2> $O/e6.err; echo "speechlike rc=$?" >> $O/rc.txt
find $O -name "*kernel_trace.csv" -delete; find $O -name "*agent_info.csv" -delete; find $O -name "*counter_collection.csv" -delete; find $O -name "*.csv" -size +4M -delete
cat $O/rc.txt; tail -2 $O/pytest.log; cat $O/pmc_traffic.txt $O/kernel_replay.txt
python - <<'P'
import json,glob
for f in sorted(glob.glob("gpurun_out/r05last/bench_*.json")):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); r=d.get("roofline",{})
        print(f.split("/")[-1], d["ms_per_step"], d.get("step_mode"), "roof", r.get("bound"), r.get("frac"), r.get("kernel_ms"), r.get("kernel_ms_min"), r.get("kernel_ms_rocprof"), r.get("traffic"), "eager", d.get("eager",{}).get("ms_per_step"))
    except Exception as e: print(f,"ERR",e)
P
tail -16 $O/step_timeline_graph.md
