#!/bin/bash
# round 5, fourth GPU pass: library-owned collectives (tests, forced-sharded lines, the RCCL graph + eager repro), the
# pruned library on the whole GPU suite, the e2e CLI leg with the prepared-database cache, ABI sanitizer with a real context
cd "$(dirname "$0")/../.."
O=gpurun_out/r05; mkdir -p $O; export TMPDIR=/tmp; R=$PWD
{ for m in 0 1; do LD_LIBRARY_PATH=/opt/rocm/lib timeout 120 experiments/rccl_graph/repro $m 20; echo "rc=$?"; done; } > $O/pass4_rccl_repro.log 2>&1
timeout 900 python tools/abi_sanitize.py > $O/pass4_abi_sanitize.log 2>&1; echo "rc=$?" >> $O/pass4_abi_sanitize.log
timeout 3000 python -m pytest tests -x -q -m gpu > $O/pass4_tests.log 2>&1
echo "tests rc=$?" >> $O/pass4_tests.log
for sc in strong weak; do
  for lc in 1 0; do
    QPG_LIB_COLLECTIVES=$lc QPG_BENCH_FORCE_SHARDED=1 MASTER_PORT=2955$lc timeout 600 python bench.py --gpus 1 --scaling $sc --n-db 2048 --steps 100 --warmup 10 --sharded-mixed-min-gflop 0 --check --no-vqvae --no-cold --no-e2e --no-cpu-baseline > $O/pass4_forced_sharded_${sc}_lib$lc.json 2> $O/pass4_forced_sharded_${sc}_lib$lc.err
  done
done
python bench.py --steps 20 --warmup 5 > $O/pass4_bench_default.json 2> $O/pass4_bench_default.err
tail -3 $O/pass4_tests.log; cat $O/pass4_rccl_repro.log; tail -2 $O/pass4_abi_sanitize.log
python - <<'P'
import json,glob
for f in sorted(glob.glob("gpurun_out/r05/pass4_forced_sharded_*.json"))+["gpurun_out/r05/pass4_bench_default.json"]:
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        print(f.split("/")[-1], d["ms_per_step"], d["step_mode"], d.get("check"), d.get("collectives",{}).get("transport","")[:30], d.get("eager",{}).get("ms_per_step"), d.get("e2e_cli"))
    except Exception as e:
        print(f, "ERR", e)
P
