#!/bin/bash
# tools/isa.sh [kernel-regex] : gfx950 assembly of hr_api.hip (device only) -> /tmp/hr.s, the matching kernel -> /tmp/k.s, plus the resource remarks
cd "$(dirname "$0")/../hanamaru-renderer_amd"
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -fno-hip-fp32-correctly-rounded-divide-sqrt -mllvm -disable-promote-alloca-to-lds -fno-slp-vectorize \
  -I../include -Ihost -Icsrc -S --offload-device-only -Rpass-analysis=kernel-resource-usage -o /tmp/hr.s csrc/hr_api.hip 2> /tmp/hr_remarks.txt
pat=${1:-_Z12trace_kernelILb0ELi5ELb1ELb0ELb0EE}
awk -v pat="$pat" '$0 ~ "^"pat && /:/ {on=1} on {print} on && /s_endpgm/ {exit}' /tmp/hr.s > /tmp/k.s
wc -l /tmp/k.s
