cd "$(dirname "$0")/../.."
export TMPDIR=/tmp; R=$PWD; O=gpurun_out/r05h; mkdir -p $O
for v in 0 1 0 1; do QPG_GATE_DEDUP=$v python tools/step_loop.py 200 graph 2>&1 | tail -1 | sed "s/^/dedup_from=$v clip1 /"; done
for v in 0 1; do QPG_GATE_DEDUP=$v QPG_LOOP_CLIPS=16 QPG_LOOP_F16=1 python tools/step_loop.py 40 graph 2>&1 | tail -1 | sed "s/^/dedup_from=$v clips16 /"; done
( cd /tmp && QPG_GATE_DEDUP=1 timeout 600 rocprofv3 --kernel-trace --output-format csv -d $R/$O/tl -- python $R/tools/step_loop.py 30 graph > $R/$O/tl.log 2>&1 )
python tools/step_timeline.py $O/tl 30 2>&1 | tail -6
find $O -name "*.csv" -delete
