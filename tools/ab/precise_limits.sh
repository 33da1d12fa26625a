#!/bin/bash
# tools/ab/precise_limits.sh: the per-path figures behind PATH_LIMITS_PRECISE (tests/test_gpu_parity.py), at each scene's test size, samplings 1 .. 8
S=1,2,3,4,5,6,7,8
python tools/ab/precise_check.py --scenes rtcamp6_v3_1 --width 320 --height 180 --samplings $S
python tools/ab/precise_check.py --scenes cornell_mini --width 160 --height 100 --samplings $S
python tools/ab/precise_check.py --scenes spheres --width 256 --height 144 --samplings $S
python tools/ab/precise_check.py --scenes rtcamp6_v2,rtcamp5,tbf3,rtcamp6_v1 --width 192 --height 108 --samplings $S
