cd /root/repo; O=gpurun_out/prio; mkdir -p $O
for i in 1 2; do
for p in 0 -1; do
  echo "eager prio=$p: $(QPG_LOOP_PRIO=$p timeout 120 python tools/step_loop.py 300 2>&1 | tail -1)" >> $O/res.txt
  echo "graph prio=$p: $(QPG_LOOP_PRIO=$p QPG_GRAPH_PRIO=$p timeout 120 python tools/step_loop.py 300 graph 2>&1 | tail -1)" >> $O/res.txt
done; done
R=/root/repo
( cd /tmp && QPG_LOOP_PRIO=-1 QPG_GRAPH_PRIO=-1 timeout 300 rocprofv3 --kernel-trace --output-format csv -d $R/$O/tl -- python $R/tools/step_loop.py 30 graph > $R/$O/tl.log 2>&1 )
python tools/step_timeline.py $O/tl 30 > $O/timeline_graph_prio.md 2>&1
( cd /tmp && QPG_LOOP_PRIO=-1 timeout 300 rocprofv3 --kernel-trace --output-format csv -d $R/$O/tle -- python $R/tools/step_loop.py 30 > $R/$O/tle.log 2>&1 )
python tools/step_timeline.py $O/tle 30 > $O/timeline_eager_prio.md 2>&1
find $O -name "*.csv" -delete
cat $O/res.txt; cat $O/timeline_graph_prio.md; cat $O/timeline_eager_prio.md
