cd /root/repo; O=gpurun_out/modeprobe; mkdir -p $O; rm -f $O/res.txt
for i in 1 2; do
for m in MODE_AUD_TXT MODE_AUD MODE_TXT; do
  echo "graph $m: $(QPG_LOOP_MODE=$m timeout 120 python tools/step_loop.py 300 graph 2>&1 | tail -1)" >> $O/res.txt
done; done
R=/root/repo
( cd /tmp && QPG_LOOP_MODE=MODE_AUD timeout 300 rocprofv3 --kernel-trace --output-format csv -d $R/$O/tl -- python $R/tools/step_loop.py 30 graph > $R/$O/tl.log 2>&1 )
python tools/step_timeline.py $O/tl 30 > $O/timeline_aud.md 2>&1
( cd /tmp && QPG_LOOP_MODE=MODE_TXT timeout 300 rocprofv3 --kernel-trace --output-format csv -d $R/$O/tlt -- python $R/tools/step_loop.py 30 graph > $R/$O/tlt.log 2>&1 )
python tools/step_timeline.py $O/tlt 30 > $O/timeline_txt.md 2>&1
find $O -name "*.csv" -delete
cat $O/res.txt; cat $O/timeline_aud.md $O/timeline_txt.md
