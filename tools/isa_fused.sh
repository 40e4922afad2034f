#!/bin/bash
# static ISA digest of one eds_fused6_kernel instantiation (default: the headline <0,4,512,1,1>) — see tools/isa_stats.py
K=${1:-eds_fused6_kernelILi0ELi4ELi512ELi1ELi1E}
cd "$(dirname "$0")/../slam-eds_amd/csrc" || exit 1
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-function -Wno-unused-value -ffp-contract=on -mllvm -amdgpu-sched-strategy=max-ilp \
    -mllvm -amdgpu-use-amdgpu-trackers=1 $EXTRA --offload-device-only -S eds_fused.hip -o /tmp/isa/fused_new.s 2>&1 | grep -E "error" 
python ../../tools/isa_stats.py x "$K" --asm /tmp/isa/fused_new.s --dump /tmp/isa/k4n.s
