#!/bin/bash
# Experiment builds of the Winograd conv kernel -> tools/dbg/libfemasr_hip_<tag>.so  (FEMASR_SO=... python tools/bench_conv.py ... --wino)
#   tt       per-phase cycle stamps (-DFEMASR_WINO_TT, both Winograd kernels): bench_conv prints the per-wave cycle shares
#   fastact  hardware exp2 / rcp SiLU in the staging (-DFEMASR_WINO_FASTACT): what the exact SiLU costs
#   ablN     -DFEMASR_WINO_ABL=N: parts of the kernel removed (bit list in kernels_wino.hip); results are garbage, timings only
#   deep0    -DFEMASR_WINO_DEEP=0: one patch register set (the round-3 prefetch distance) for A/B against the default two-set form
set -e
cd "$(dirname "$0")/.."; mkdir -p tools/dbg
F="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt -Wno-unused-result"
O=femasr_amd/csrc
for tag in ${@:-tt}; do
  case $tag in tt) D="-DFEMASR_WINO_TT=1 $EXTRA_DEFS";; abl*) D="-DFEMASR_WINO_ABL=${tag#abl}";; deep*) D="-DFEMASR_WINO_DEEP=${tag#deep}";; nt*) D="-DFEMASR_WINO_NT=${tag#nt}";; *) D="$EXTRA_DEFS";; esac
  /opt/rocm/bin/hipcc $F $D -c $O/kernels_wino.hip -o tools/dbg/kernels_wino_$tag.o
  /opt/rocm/bin/hipcc $F $D -c $O/kernels_wino_up2.hip -o tools/dbg/kernels_wino_up2_$tag.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o tools/dbg/libfemasr_hip_$tag.so $O/kernels_conv.o $O/kernels_gemm.o $O/kernels_gemm_bf16.o $O/kernels_vq.o tools/dbg/kernels_wino_$tag.o tools/dbg/kernels_wino_up2_$tag.o $O/kernels_conv_bf16.o $O/kernels_misc.o $O/model.o
done
