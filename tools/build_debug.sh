#!/bin/bash
# Debug build with per-tap cycle stamps in the bf16x3 conv kernel (-DFEMASR_TAPTIME) -> tools/dbg/libfemasr_hip_tt.so
set -e
cd "$(dirname "$0")/.."; mkdir -p tools/dbg
F="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt -Wno-unused-result"
/opt/rocm/bin/hipcc $F -DFEMASR_TAPTIME=${TT_LEVEL:-1} -c femasr_amd/csrc/kernels_conv_bf16.hip -o tools/dbg/kernels_conv_bf16_tt.o
/opt/rocm/bin/hipcc $F -DFEMASR_TAPTIME=${TT_LEVEL:-1} -c femasr_amd/csrc/kernels_conv.hip -o tools/dbg/kernels_conv_tt.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o tools/dbg/libfemasr_hip_tt.so tools/dbg/kernels_conv_tt.o tools/dbg/kernels_conv_bf16_tt.o femasr_amd/csrc/kernels_misc.o femasr_amd/csrc/model.o
