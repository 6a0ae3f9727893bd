cd /root/repo
L=experiments/audio_hl
for r in 1 2 3 4 5; do for v in nt0 nt1; do QPG_LIB_PATH=$L/libqpg_p$v.so timeout 300 python bench.py --steps 200 --warmup 10 --no-vqvae --no-cold --no-e2e --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); r = d['roofline']
print('bench $v: step %.4f ms  sweep mean %.4f min %.4f median %.4f  eager %.4f pipelined %s' % (d['ms_per_step'], r['kernel_ms'], r['kernel_ms_min'], r['kernel_ms_median'], d.get('eager', {}).get('ms_per_step', 0), (d.get('pipelined') or {}).get('ms_per_clip')))"; done; done
