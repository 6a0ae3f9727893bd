#!/bin/bash
# Round-end GPU pass: GPU test suite, build()+smoke(), default bench line, rocprofv3 kernel stats of the same command,
# FETCH_SIZE / WRITE_SIZE passes on the audio sweep.  Everything lands under gpurun_out/final/.
set -u
O=gpurun_out/final; mkdir -p $O
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" > $O/rc.txt
timeout 600 python -c "import __graft_entry__ as g; g.build(); g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1; echo "smoke rc=$?" >> $O/rc.txt
timeout 900 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc=$?" >> $O/rc.txt
R=$PWD
( cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof -o bench -- python $R/bench.py > $R/$O/bench_profiled.json 2> $R/$O/prof.err ); echo "prof rc=$?" >> $O/rc.txt
for c in FETCH_SIZE WRITE_SIZE; do
  for mode in f64 mx; do
    ( cd /tmp && timeout 600 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $R/$O/pmc_${c}_$mode -o a -- python $R/tools/bench_audio.py 2048 48 3 $mode > $R/$O/pmc_${c}_$mode.log 2>&1 ); echo "pmc $c $mode rc=$?" >> $O/rc.txt
    python tools/pmc_summary.py $O/pmc_${c}_$mode audio > $O/pmc_${c}_$mode.txt 2>&1
  done
done
find $O -name "*.csv" -size +8M -delete; find $O -name "*kernel_trace.csv" -delete
cat $O/rc.txt; tail -3 $O/pytest.log; cat $O/bench.json
