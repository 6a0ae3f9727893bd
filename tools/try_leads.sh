python -m pytest tests/test_gpu_matching.py tests/test_gpu_text_prefilter.py -q -m gpu -x 2>&1 | tail -2
for l in 0 0.1 0.2 0.3 0.45; do echo "lead $l"; QPG_BENCH_TEXT_LEAD=$l python bench.py --no-f64-line 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['roofline']['kernel_ms'])"; done
