cd /root/repo; O=gpurun_out/cut2; mkdir -p $O; rm -f $O/res.txt
timeout 1500 python -m pytest tests -x -q -m gpu > $O/pytest.log 2>&1; echo "pytest rc=$? $(tail -1 $O/pytest.log)" >> $O/res.txt
timeout 600 python tools/stress_parity.py 40 > $O/parity.log 2>&1; echo "parity rc=$? $(tail -1 $O/parity.log)" >> $O/res.txt
timeout 600 python tools/stress_mixed.py 30 > $O/mixed.log 2>&1; echo "mixed rc=$? $(tail -1 $O/mixed.log)" >> $O/res.txt
for c in 0 1 0 1; do
  QPG_RANK_CUT=$c timeout 600 python bench.py --steps 200 --warmup 5 --no-cpu-baseline --no-vqvae --no-e2e --no-cold > $O/bench_cut$c.json 2> $O/bench_cut$c.err
  python - <<PY >> $O/res.txt
import json; d=json.load(open("$O/bench_cut$c.json")); print("cut=$c", d["ms_per_step"], d["value"], d["roofline"]["kernel_ms"], d["eager"]["ms_per_step"], d["eager"]["tier1_pairs_per_step"], d.get("mixed_precision",{}).get("codes_equal_f64_sweep"), d["pipelined"]["ms_per_step"])
PY
done
cat $O/res.txt
