#!/bin/bash
# tools/ab/prec_variants.sh libA libB ...: precise shading in the megakernel (mode 3) against fp32 (mode 0), library by library on one box
cp hanamaru-renderer_amd/libhanamaru_hip.so /tmp/lib_cur.so
for L in "$@"; do
  cp $L hanamaru-renderer_amd/libhanamaru_hip.so
  echo "== $L"
  python tools/ab/split_ab.py --modes ${MODES:-0,3} --scenes ${SCENES:-simple,cornell_mini,rtcamp6_v3_1,tbf3} --samplings 32 2>/dev/null | cut -c1-120
done
cp /tmp/lib_cur.so hanamaru-renderer_amd/libhanamaru_hip.so
