# FETCH_SIZE / WRITE_SIZE of the audio sweeps (one counter per pass, --kernel-trace only) -> gpurun_out/nt
set -u
export TMPDIR=/tmp; R=$PWD; O=gpurun_out/nt; mkdir -p $O
for c in FETCH_SIZE WRITE_SIZE; do
  ( cd /tmp && timeout 600 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $R/$O/pmc_$c -o a -- python $R/tools/bench_audio_hl.py > $R/$O/pmc_$c.log 2>&1 )
  python tools/pmc_summary.py $O/pmc_$c audio_cosine_hl
done
grep "hl sweep" $O/pmc_WRITE_SIZE.log
find $O -name "*.csv" -delete
