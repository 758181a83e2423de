#!/bin/bash
# tools/mkvariant.sh NAME [-DSWITCH ...] : build dd2360-raytracing_amd/variants/lib_NAME.so (the product library with extra -D switches) for tools/ab.sh
set -e
name=$1; shift
root=$(cd "$(dirname "$0")/.." && pwd)/dd2360-raytracing_amd
obj=$(mktemp -d /tmp/rt_variant_XXXX)      # objects outside the tree: only the finished library travels with gpurun
mkdir -p $root/variants
F="--offload-arch=gfx950 -std=c++17 -O3 -ffp-contract=off -fno-fast-math -fno-gpu-flush-denormals-to-zero -fPIC -Wall -Wno-unused-function $*"
cd $root
/opt/rocm/bin/hipcc $F -fno-slp-vectorize -DRT_SPLIT_LIST -c -o $obj/k.o csrc/rt_kernels.hip &
/opt/rocm/bin/hipcc $F -DRT_SPLIT_LIST -c -o $obj/l.o csrc/rt_kernels_list.hip &
/opt/rocm/bin/hipcc ${F/-ffp-contract=off/-ffp-contract=fast} -fno-slp-vectorize -c -o $obj/c.o csrc/rt_kernels_contract.hip &
/opt/rocm/bin/hipcc $F -c -o $obj/h.o csrc/rt_kernels_fp16.hip &
/opt/rocm/bin/hipcc $F -c -o $obj/a.o csrc/rt_api.hip &
/opt/rocm/bin/hipcc $F -c -o $obj/m.o csrc/rt_multi.hip &
/opt/rocm/bin/hipcc $F -c -o $obj/b.o csrc/rt_build.hip &
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o variants/lib_$name.so $obj/k.o $obj/l.o $obj/c.o $obj/h.o $obj/a.o $obj/m.o $obj/b.o -ldl
rm -rf $obj
echo "built variants/lib_$name.so"
