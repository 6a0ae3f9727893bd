for rep in 1 2 3; do for p in 64 32 16 48; do echo -n "probe $p rep $rep: "; QPG_LOOP_PROBE=$p python tools/step_loop.py 400 graph 2>/dev/null | tail -1; done; done
