cd /root/repo
export TMPDIR=/tmp; R=$PWD; O=gpurun_out/r05nt; mkdir -p $O
for gp in 0 2 20; do
  rm -rf $O/tlg
  ( cd /tmp && QPG_LOOP_GAP_MS=$gp QPG_LOOP_MODE=MODE_AUD timeout 600 rocprofv3 --kernel-trace --output-format csv -d $R/$O/tlg -- python $R/tools/step_loop.py 60 graph > $R/$O/tlg.log 2>&1 )
  echo "== gap $gp ms"; python tools/step_timeline.py $O/tlg 60 2>&1 | grep "audio_cosine_hl2\|mixed_stream\|GPU-side span"
done
find $O -name "*.csv" -delete
