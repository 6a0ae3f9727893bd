# FETCH_SIZE / WRITE_SIZE of EVERY kernel of the matching step (tools/step_loop.py), one counter per pass -> gpurun_out/pmcstep
set -u
export TMPDIR=/tmp; R=$PWD; O=gpurun_out/pmcstep; mkdir -p $O
for c in FETCH_SIZE WRITE_SIZE; do
  ( cd /tmp && timeout 600 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $R/$O/pmc_$c -o a -- python $R/tools/step_loop.py 20 > $R/$O/pmc_$c.log 2>&1 )
  python tools/pmc_summary.py $O/pmc_$c > $O/$c.txt 2>&1
done
find $O -name "*.csv" -delete
cat $O/FETCH_SIZE.txt; cat $O/WRITE_SIZE.txt
