cd /root/repo; O=gpurun_out/gate; mkdir -p $O; rm -f $O/res.txt
for i in 1 2 3; do
for g in none stream; do
  echo "graph gate=$g: $(QPG_TEXT_GATE=$g timeout 120 python tools/step_loop.py 300 graph 2>&1 | tail -1)" >> $O/res.txt
done; done
R=/root/repo
( cd /tmp && QPG_TEXT_GATE=stream timeout 300 rocprofv3 --kernel-trace --output-format csv -d $R/$O/tl -- python $R/tools/step_loop.py 30 graph > $R/$O/tl.log 2>&1 )
python tools/step_timeline.py $O/tl 30 > $O/timeline_gate.md 2>&1
find $O -name "*.csv" -delete
QPG_TEXT_GATE=stream timeout 900 python -m pytest tests/test_gpu_matching.py -x -q -m gpu 2>&1 | tail -2 > $O/pytest.txt
cat $O/res.txt; cat $O/timeline_gate.md; cat $O/pytest.txt
