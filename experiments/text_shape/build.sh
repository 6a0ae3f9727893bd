#!/bin/bash
# builds libtext_<QB>_<NG>.so: qpg_text.hip with the Q > 24 launch shape replaced (run from the repo root)
for v in "12,4" "6,4" "6,8" "4,4" "3,8" "4,12"; do
  qb=${v%,*}; ng=${v#*,}
  sed "s|if (Q > 24) return launch_text<12, 4>|if (Q > 24) return launch_text<$qb, $ng>|" qpgesture_amd/csrc/qpg_text.hip > /tmp/qpg_text_v.hip
  /opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -fPIC -shared -I include -I qpgesture_amd/csrc \
    -o experiments/text_shape/libtext_${qb}_${ng}.so /tmp/qpg_text_v.hip qpgesture_amd/csrc/qpg_core.hip
done
