#!/bin/bash
# tools/ab/mkvariant.sh <name> [extra hipcc flags]: the HIP library of the tree as it stands -> tools/ab/lib_<name>.so (git-ignored; travels to the GPU box)
set -e
cd "$(dirname "$0")/../../hanamaru-renderer_amd"
make -s csrc/bvh_build.o csrc/flatten.o
name=$1; shift
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -fno-hip-fp32-correctly-rounded-divide-sqrt -mllvm -disable-promote-alloca-to-lds -fno-slp-vectorize \
  -I../include -Ihost -Icsrc "$@" -Rpass-analysis=kernel-resource-usage -c -o /tmp/var_$name.o csrc/hr_api.hip 2> /tmp/res_$name.txt
/opt/rocm/bin/hipcc --offload-arch=gfx950 -fPIC -shared -o ../tools/ab/lib_$name.so /tmp/var_$name.o csrc/bvh_build.o csrc/flatten.o -ldl
python3 ../tools/kres.py /tmp/res_$name.txt /tmp/res_$name.txt "trace_kernelILb0ELi5ELb1ELb0ELb0" | grep -v "^ *->"
