#!/bin/bash
# Device-only assembly of conv.hip -> gpurun_out/dis/conv.s, then prints register use and part of one kernel
# instantiation: tools/dis_conv.sh <mangled template suffix> <kernel name> <n-th s_barrier to start at> <lines>
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out/dis
(cd wsi_segmentation_pipeline_amd/csrc && hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -S --cuda-device-only -o ../../gpurun_out/dis/conv.s conv.hip 2>&1 | grep -v "warning\|^$" | tail -3)
K=${1:-ILi4ELi1ELi4ELi3ELi2ELb1EE}; NAME=${2:-conv3x3s1_slab3_kernel}
awk -v k="$NAME$K" '$0 ~ "\\.name:.*"k {f=1} f&&/vgpr_count|vgpr_spill/{print k, $0} f&&/wavefront_size/{exit}' gpurun_out/dis/conv.s
a=$(grep -n "^_Z[0-9]*${NAME}${K}v.*:" gpurun_out/dis/conv.s | head -1 | cut -d: -f1)
sed -n "${a},$((a+4000))p" gpurun_out/dis/conv.s > gpurun_out/dis/k.s
b=$(grep -n "s_barrier" gpurun_out/dis/k.s | sed -n "${3:-1}p" | cut -d: -f1)
sed -n "${b},$((b+${4:-100}))p" gpurun_out/dis/k.s | cut -c1-110 | grep -v "^\s*;"
