#!/bin/bash
# Samples the GPU's clocks and power while a command runs:  tools/powerwatch.sh <outfile> <command...>
OUT=$1; shift
( while true; do rocm-smi --showclocks --showpower --showtemp 2>/dev/null | grep -E "sclk|mclk|fclk|Power|Temperature \(Sensor (edge|junction)" | tr '\n' ' ' ; echo; sleep 0.5; done ) > $OUT 2>&1 &
W=$!
"$@"
kill $W
