#!/bin/bash
# tools/ab/abscenes.sh "<scenes>" libA libB ... : abn.sh over several scenes
scenes="$1"; shift
for sc in $scenes; do echo "== $sc"; BENCH_ARGS="--scene $sc ${BENCH_ARGS_EXTRA:-}" tools/ab/abn.sh "$@"; done
