#!/bin/bash
# round 5, last pass: the whole GPU suite on the final tree, the default line, the multi-clip lines (text side of >= 256 queries
# on the by-code path since the evidence pass), cfg-3 with its traffic field read from the committed PMC summary
cd "$(dirname "$0")/../.."
O=gpurun_out/r05final; mkdir -p $O; export TMPDIR=/tmp; R=$PWD
timeout 3000 python -m pytest tests -m gpu -q > $O/pytest.log 2>&1; echo "pytest rc=$?" > $O/rc.txt
timeout 600 python -c "import __graft_entry__ as g; g.build(); g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1; echo "smoke rc=$?" >> $O/rc.txt
timeout 900 python bench.py --steps 20 --warmup 5 > $O/bench_20.json 2> $O/bench_20.err; echo "bench20 rc=$?" >> $O/rc.txt
( cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof -o bench -- python $R/bench.py --steps 20 --warmup 5 > $R/$O/bench_profiled.json 2> $R/$O/prof.err ); echo "prof rc=$?" >> $O/rc.txt
python tools/make_profile_summary.py $O/prof $O/bench_n1 "python bench.py --steps 20 --warmup 5 (N=1) under rocprofv3 --kernel-trace --stats" > /dev/null 2>&1
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --output-format csv -d $R/$O/tlg -- python $R/tools/step_loop.py 30 graph > $R/$O/tlg.log 2>&1 )
python tools/step_timeline.py $O/tlg 30 > $O/step_timeline_graph.md 2>&1
timeout 900 python bench.py --steps 30 --warmup 5 --clips 16 --feature-dtype f16 --encode-batch 96 --no-vqvae --no-cold --no-e2e --no-cpu-baseline > $O/bench_clips16_f16_enc96.json 2> $O/e1.err; echo "c16 enc rc=$?" >> $O/rc.txt
timeout 900 python bench.py --steps 30 --warmup 5 --clips 16 --feature-dtype f16 --encode-batch 96 --encode-precision f16x3 --no-vqvae --no-cold --no-e2e --no-cpu-baseline > $O/bench_clips16_f16_enc96_f16x3.json 2> $O/e2.err; echo "c16 enc f16x3 rc=$?" >> $O/rc.txt
timeout 900 python bench.py --steps 30 --warmup 5 --clips 16 --feature-dtype f16 --no-vqvae --no-cold --no-e2e --no-cpu-baseline > $O/bench_clips16_f16.json 2> $O/e3.err; echo "c16 rc=$?" >> $O/rc.txt
timeout 900 python bench.py --steps 30 --warmup 5 --clips 16 --no-vqvae --no-cold --no-e2e --no-cpu-baseline > $O/bench_clips16_f32.json 2> $O/e4.err; echo "c16 f32 rc=$?" >> $O/rc.txt
timeout 900 python bench.py --workload cfg3 > $O/bench_cfg3.json 2> $O/e5.err; echo "cfg3 rc=$?" >> $O/rc.txt
( cd /tmp && QPG_LOOP_CLIPS=16 QPG_LOOP_F16=1 QPG_LOOP_ENC=96 timeout 600 rocprofv3 --kernel-trace --output-format csv -d $R/$O/tl16 -- python $R/tools/step_loop.py 20 graph > $R/$O/tl16.log 2>&1 )
python tools/step_timeline.py $O/tl16 20 > $O/step_timeline_c16_f16_enc96_graph.md 2>&1
( cd /tmp && QPG_LOOP_CLIPS=16 QPG_LOOP_F16=1 timeout 600 rocprofv3 --kernel-trace --output-format csv -d $R/$O/tl16b -- python $R/tools/step_loop.py 20 graph > $R/$O/tl16b.log 2>&1 )
python tools/step_timeline.py $O/tl16b 20 > $O/step_timeline_c16_f16_graph.md 2>&1
find $O -name "*kernel_trace.csv" -delete; find $O -name "*agent_info.csv" -delete; find $O -name "*.csv" -size +4M -delete
cat $O/rc.txt; tail -2 $O/pytest.log
python - <<'P'
import json,glob
for f in sorted(glob.glob("gpurun_out/r05final/bench_*.json")):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); r=d.get("roofline",{})
        print(f.split("/")[-1], d["ms_per_step"], d.get("step_mode"), "roof", r.get("bound"), r.get("frac"), r.get("kernel_ms"), r.get("kernel_ms_rocprof"), r.get("traffic"), "eager", d.get("eager",{}).get("ms_per_step"))
    except Exception as e: print(f,"ERR",e)
P
