#!/bin/bash
# tools/regs.sh [extra -D flags]: register use of the render kernels (hipcc -Rpass-analysis=kernel-resource-usage on csrc/rt_kernels.hip and rt_kernels_fp16.hip)
cd "$(dirname "$0")/../dd2360-raytracing_amd"
F="--offload-arch=gfx950 -std=c++17 -O3 -ffp-contract=off -fno-fast-math -fno-gpu-flush-denormals-to-zero -fPIC -Wno-unused-function $*"
for f in rt_kernels rt_kernels_fp16; do
  /opt/rocm/bin/hipcc $F -fno-slp-vectorize -DRT_SPLIT_LIST -c --cuda-device-only -o /dev/null csrc/$f.hip -Rpass-analysis=kernel-resource-usage 2>&1 | grep "remark:" |
  awk '/Function Name:/{n=$5} / VGPRs:/{v=$4} /TotalSGPRs:/{s=$4} /ScratchSize/{sc=$5} /VGPRs Spill/{vs=$5} /SGPRs Spill/{ss=$5} /Occupancy/{o=$5} /LDS Size/{print n, "VGPR", v, "SGPR", s, "scratch", sc, "vspill", vs, "sspill", ss, "occ", o}' | c++filt | grep -E "k_render<|k_render_h<|k_tile_cost" | sed 's/(rt::RenderArgs.*)//'
done
