#!/bin/bash
# Debug build of the Winograd conv kernel with per-phase cycle stamps (-DFEMASR_WINO_TT) -> tools/dbg/libfemasr_hip_tt.so
# (python tools/bench_conv.py ... --wino with FEMASR_SO=tools/dbg/libfemasr_hip_tt.so prints the per-wave cycle shares)
set -e
cd "$(dirname "$0")/.."; mkdir -p tools/dbg
F="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt -Wno-unused-result"
/opt/rocm/bin/hipcc $F -DFEMASR_WINO_TT=1 -c femasr_amd/csrc/kernels_wino.hip -o tools/dbg/kernels_wino_tt.o
O=femasr_amd/csrc
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o tools/dbg/libfemasr_hip_tt.so $O/kernels_conv.o $O/kernels_gemm.o $O/kernels_vq.o tools/dbg/kernels_wino_tt.o $O/kernels_conv_bf16.o $O/kernels_misc.o $O/model.o
