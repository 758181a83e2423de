#!/bin/bash
# tools/mkvariant.sh NAME [-DSWITCH ...] : build dd2360-raytracing_amd/variants/lib_NAME.so (the product library with extra -D switches) for tools/ab.sh
set -e
name=$1; shift
root=$(cd "$(dirname "$0")/.." && pwd)/dd2360-raytracing_amd
mkdir -p $root/variants $root/build/v_$name
F="--offload-arch=gfx950 -std=c++17 -O3 -ffp-contract=off -fno-fast-math -fno-gpu-flush-denormals-to-zero -fPIC -Wall -Wno-unused-function $*"
cd $root
/opt/rocm/bin/hipcc $F -fno-slp-vectorize -DRT_SPLIT_LIST -c -o build/v_$name/k.o csrc/rt_kernels.hip &
/opt/rocm/bin/hipcc $F -DRT_SPLIT_LIST -c -o build/v_$name/l.o csrc/rt_kernels_list.hip &
/opt/rocm/bin/hipcc $F -c -o build/v_$name/h.o csrc/rt_kernels_fp16.hip &
/opt/rocm/bin/hipcc $F -c -o build/v_$name/a.o csrc/rt_api.hip &
/opt/rocm/bin/hipcc $F -c -o build/v_$name/m.o csrc/rt_multi.hip &
/opt/rocm/bin/hipcc $F -c -o build/v_$name/b.o csrc/rt_build.hip &
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o variants/lib_$name.so build/v_$name/k.o build/v_$name/l.o build/v_$name/h.o build/v_$name/a.o build/v_$name/m.o build/v_$name/b.o -ldl
echo "built variants/lib_$name.so"
