#!/bin/bash
cp hanamaru-renderer_amd/libhanamaru_hip.so /tmp/l.so
for sc in rtcamp6_v2 rtcamp6_v1 tbf3 rtcamp6_v3_1 rtcamp5; do for k in "tail_div=0" "tail_div=16" "tail_div=8" "tail_div=0" "tail_div=16" "tail_div=32"; do
BENCH_ARGS="--scene $sc --debug $k" tools/ab/abn2.sh /tmp/l.so 2>&1 | head -1 | sed "s/^/$sc $k /"
done; done
