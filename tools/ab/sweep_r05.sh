#!/bin/bash
cp hanamaru-renderer_amd/libhanamaru_hip.so /tmp/l.so
for k in "trace_budget=0" "trace_budget=1536" "trace_budget=896" "trace_budget=768" "trace_budget=0" "trace_budget=1536"; do
BENCH_ARGS="--scene rtcamp6_dodeca --width 3840 --height 2160 --spp-per-step 4 --debug $k" tools/ab/abn2.sh /tmp/l.so 2>&1 | head -1 | sed "s/^/dodeca4k $k /"
done
for sc in rtcamp5 tbf3 rtcamp6_dodeca; do for k in "trace_budget=0" "trace_budget=1536" "trace_budget=0" "trace_budget=1536"; do
BENCH_ARGS="--scene $sc --debug $k" tools/ab/abn2.sh /tmp/l.so 2>&1 | head -1 | sed "s/^/$sc $k /"
done; done
