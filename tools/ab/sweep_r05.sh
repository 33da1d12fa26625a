#!/bin/bash
cp hanamaru-renderer_amd/libhanamaru_hip.so /tmp/l.so
for sc in rtcamp6_v2 rtcamp6_v1; do for k in "" "--split-ratio 0" "--split-ratio 0.5" "--split-ratio 1" "--split-ratio 2" "--split-ratio 4" "--max-leaf 2" "--max-leaf 3" "--max-leaf 6" "--bvh-builder 2"; do
BENCH_ARGS="--scene $sc $k" tools/ab/abn.sh /tmp/l.so 2>&1 | head -1 | sed "s/^/$sc $k /"
done; done
