cd /root/repo; O=gpurun_out/cut; mkdir -p $O; rm -f $O/res.txt
timeout 900 python -m pytest tests/test_gpu_fullsize.py -x -q -m gpu -k "relevance_cut" -s 2>&1 | tail -15 > $O/pytest_cut.txt
for i in 1 2; do
for c in 0 1; do
  echo "graph cut=$c: $(QPG_RANK_CUT=$c timeout 120 python tools/step_loop.py 300 graph 2>&1 | tail -1)" >> $O/res.txt
done; done
R=/root/repo
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --output-format csv -d $R/$O/tl -- python $R/tools/step_loop.py 30 graph > $R/$O/tl.log 2>&1 )
python tools/step_timeline.py $O/tl 30 > $O/timeline_cut.md 2>&1
find $O -name "*.csv" -delete
timeout 1200 python -m pytest tests/test_gpu_matching.py tests/test_gpu_mixed.py tests/test_gpu_guard_overflow.py -x -q -m gpu 2>&1 | tail -4 > $O/pytest_rest.txt
cat $O/pytest_cut.txt $O/res.txt $O/timeline_cut.md $O/pytest_rest.txt
