for ta in 1 0; do for af in 0 1; do echo "TEXT_AFTER=$ta AUDIO_FIRST=$af"; QPG_TEXT_AFTER=$ta QPG_AUDIO_FIRST=$af python tools/step_loop.py 300 2>&1 | tail -1; done; done
