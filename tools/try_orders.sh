# bench.py under the three schedules of the text side, alternating (box-to-box variance is ~5 %: compare within a run)
#   after: behind the audio sweep; --text-first: enqueued first on its own stream; QPG_BENCH_AUDIO_FIRST=1: behind the
#   audio side's launches, no ordering
for rep in 1 2 3; do
  for mode in "after" "text-first" "audio-first"; do
    fl=""; ev=""
    [ $mode = text-first ] && fl="--text-first"
    [ $mode = audio-first ] && ev="QPG_BENCH_AUDIO_FIRST=1"
    [ $mode = after ] && ev="QPG_BENCH_AUDIO_FIRST=0"
    env $ev python bench.py --no-f64-line --steps 400 $fl 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$mode', d['ms_per_step'], d['roofline']['kernel_ms'])"
  done
done
