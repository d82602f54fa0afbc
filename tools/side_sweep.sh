#!/bin/bash
# dev tool: the 4K frame (and 1080p / 1600) with the shadow rays of levels 0..k-1 on the side stream, for k = 1..6
for L in 1 2 3 4 6; do echo "== side stream on at every size, RT_SHADOW_SIDE_LEVELS=$L"; RT_SHADOW_SIDE_SAMPLES=4000000000 RT_SHADOW_SIDE_LEVELS=$L timeout -k 10 200 python tools/small_frames.py 2>/dev/null | grep -E "part 0/1|part 0/8|part 0/2" | head -6; done
