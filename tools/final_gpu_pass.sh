#!/bin/bash
# GPU evidence pass: GPU test suite, build()+smoke(), default bench line, rocprofv3 kernel stats of the same command, the
# step timeline, FETCH_SIZE / WRITE_SIZE passes on the audio sweeps (counters in their own runs, --kernel-trace only).
# Everything lands under gpurun_out/final/; usage: tools/final_gpu_pass.sh [tag]
set -u
O=gpurun_out/final; mkdir -p $O
export TMPDIR=/tmp
R=$PWD
timeout 1800 python -m pytest tests -m gpu -q > $O/pytest.log 2>&1; echo "pytest rc=$?" > $O/rc.txt
timeout 600 python -c "import __graft_entry__ as g; g.build(); g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1; echo "smoke rc=$?" >> $O/rc.txt
timeout 900 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc=$?" >> $O/rc.txt
timeout 900 python bench.py --steps 20 --warmup 5 > $O/bench_20.json 2> $O/bench_20.err; echo "bench20 rc=$?" >> $O/rc.txt
( cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof -o bench -- python $R/bench.py > $R/$O/bench_profiled.json 2> $R/$O/prof.err ); echo "prof rc=$?" >> $O/rc.txt
python tools/make_profile_summary.py $O/prof $O/bench_n1 "python bench.py (N=1, 200 steps) under rocprofv3 --kernel-trace --stats" > /dev/null 2>&1
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $R/$O/tl -- python $R/tools/step_loop.py 30 > $R/$O/tl.log 2>&1 )
python tools/step_timeline.py $O/tl 30 > $O/step_timeline.md 2>&1
timeout 900 python bench.py --clips 16 > $O/bench_clips16.json 2> $O/bench_clips16.err; echo "clips16 rc=$?" >> $O/rc.txt
timeout 900 python bench.py --scaling strong > $O/bench_strong.json 2> $O/bench_strong.err; echo "strong rc=$?" >> $O/rc.txt
timeout 900 python bench.py --workload cfg3 > $O/bench_cfg3.json 2> $O/bench_cfg3.err; echo "cfg3 rc=$?" >> $O/rc.txt
timeout 900 python bench.py --data speechlike > $O/bench_speechlike.json 2> $O/bench_speechlike.err; echo "speechlike rc=$?" >> $O/rc.txt
timeout 300 python tools/bench_train.py 256 > $O/bench_train.log 2>&1; echo "bench_train rc=$?" >> $O/rc.txt
timeout 300 python tools/prof_train_layers.py > $O/train_layers.md 2>&1; echo "train_layers rc=$?" >> $O/rc.txt
for c in FETCH_SIZE WRITE_SIZE; do
  ( cd /tmp && timeout 600 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $R/$O/pmc_$c -o a -- python $R/tools/bench_audio_hl.py > $R/$O/pmc_$c.log 2>&1 ); echo "pmc $c rc=$?" >> $O/rc.txt
  python tools/pmc_summary.py $O/pmc_$c audio > $O/pmc_$c.txt 2>&1
done
find $O -name "*.csv" -size +8M -delete; find $O -name "*kernel_trace.csv" -delete; find $O -name "*memory_copy_trace.csv" -delete
cat $O/rc.txt; tail -3 $O/pytest.log; head -c 600 $O/bench.json; echo; cat $O/step_timeline.md | tail -16; cat $O/pmc_*.txt; grep forward $O/bench_train.log
