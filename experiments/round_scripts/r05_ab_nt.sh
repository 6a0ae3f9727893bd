#!/bin/bash
# non-temporal fragment loads in the sweeps (H2_NT: 0 never, 1 one query chunk only = product, 2 always), alternating on one
# box: the kernels alone (30 back-to-back launches), then bench.py's step with the sweep bracketed by HIP events
cd "$(dirname "$0")/../.."
O=gpurun_out/r05nt; mkdir -p $O; L=experiments/audio_hl
{
for r in 1 2 3; do for v in nt0 nt1; do echo "== Q=48 $v"; QPG_LIB_PATH=$L/libqpg_p$v.so timeout 300 python tools/bench_audio_hl.py 2048 48 2>&1 | grep "hl sweep\|hl1" | sed 's/|  *mx.*//'; done; done
for r in 1 2; do for v in nt0 nt2; do echo "== Q=768 $v"; QPG_LIB_PATH=$L/libqpg_p$v.so timeout 300 python tools/bench_audio_hl.py 2048 768 2>&1 | grep "hl1"; done; done
for r in 1 2 3; do for v in nt0 nt1; do QPG_LIB_PATH=$L/libqpg_p$v.so timeout 300 python bench.py --steps 200 --warmup 10 --no-vqvae --no-cold --no-e2e --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); r = d['roofline']
print('bench $v: step %.4f ms  sweep mean %.4f min %.4f median %.4f  eager %.4f  check %s' % (d['ms_per_step'], r['kernel_ms'], r['kernel_ms_min'], r['kernel_ms_median'], d.get('eager', {}).get('ms_per_step', 0), d.get('codes_equal_f64_sweep')))"; done; done
for r in 1 2; do for v in nt0 nt1; do QPG_LIB_PATH=$L/libqpg_p$v.so timeout 300 python bench.py --steps 100 --warmup 10 --feature-dtype f16 --no-vqvae --no-cold --no-e2e --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); r = d['roofline']
print('bench f16 track $v: step %.4f ms  sweep mean %.4f min %.4f median %.4f' % (d['ms_per_step'], r['kernel_ms'], r['kernel_ms_min'], r['kernel_ms_median']))"; done; done
} > $O/ab_nt.log 2>&1
cat $O/ab_nt.log
