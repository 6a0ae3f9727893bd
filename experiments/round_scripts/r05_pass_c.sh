#!/bin/bash
# round 5, pass c: the re-ordered sweep (query loads in front of the ring, prologue in the loop's order, refill pinned) and
# the cfg-3 path (h-plane prefilter + by-code select): tests, kernel timings, bench lines
cd "$(dirname "$0")/../.."
O=gpurun_out/r05c; mkdir -p $O; export TMPDIR=/tmp; R=$PWD
timeout 1500 python -m pytest tests/test_gpu_cfg3.py tests/test_gpu_audio_hl.py tests/test_gpu_text_prefilter.py -x -q -m gpu -s > $O/tests.log 2>&1; echo "tests rc=$?" >> $O/tests.log
{ python tools/bench_audio_hl.py 2048 48; python tools/bench_audio_hl.py 2048 48; python tools/bench_audio_hl.py 2048 768; } > $O/kernels.log 2>&1
timeout 600 python bench.py --workload cfg3 > $O/bench_cfg3.json 2> $O/bench_cfg3.err
timeout 600 python bench.py --steps 100 --warmup 10 --no-vqvae --no-cold --no-e2e --no-cpu-baseline > $O/bench_100.json 2> $O/bench_100.err
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof_cfg3 -o cfg3 -- python $R/bench.py --workload cfg3 > $R/$O/bench_cfg3_prof.json 2> $R/$O/prof_cfg3.err )
grep -E "passed|failed|rc=|prefilter|cfg-3" $O/tests.log | tail -12; cat $O/kernels.log | grep -v amdgpu.ids
python - <<'P'
import json
for f in ("bench_cfg3","bench_100"):
    try:
        d=json.loads(open("gpurun_out/r05c/%s.json"%f).read().strip().splitlines()[-1]); r=d["roofline"]
        print(f, d["ms_per_step"], r.get("kernel_ms"), r.get("frac"), d.get("eager",{}).get("ms_per_step"))
    except Exception as e: print(f,"ERR",e)
P
head -12 $O/prof_cfg3/*kernel_stats.csv 2>/dev/null | cut -c1-160
find $O -name "*kernel_trace.csv" -delete; find $O -name "*agent_info.csv" -delete
