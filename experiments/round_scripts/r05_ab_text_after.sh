cd /root/repo 2>/dev/null || cd $GRAFT_REPO_ROOT
for v in 0 1 0 1; do QPG_TEXT_AFTER=$v QPG_LOOP_CLIPS=16 QPG_LOOP_F16=1 python tools/step_loop.py 40 graph 2>&1 | tail -1 | sed "s/^/text_after=$v clips16 f16 /"; done
for v in 0 1; do QPG_TEXT_AFTER=$v QPG_LOOP_CLIPS=16 QPG_LOOP_F16=1 QPG_LOOP_ENC=96 python tools/step_loop.py 40 graph 2>&1 | tail -1 | sed "s/^/text_after=$v clips16 f16 enc96 /"; done
for v in 0 1; do QPG_TEXT_AFTER=$v QPG_LOOP_CLIPS=16 python tools/step_loop.py 40 graph 2>&1 | tail -1 | sed "s/^/text_after=$v clips16 f32 /"; done
for v in 0 1; do QPG_TEXT_AFTER=$v QPG_LOOP_CLIPS=4 QPG_LOOP_F16=1 python tools/step_loop.py 40 graph 2>&1 | tail -1 | sed "s/^/text_after=$v clips4 f16 /"; done
