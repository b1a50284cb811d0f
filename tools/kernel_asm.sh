#!/bin/bash
# Device assembly + register / occupancy report of one kernel source (same flags as the Makefile).
#   tools/kernel_asm.sh attention.hip [extra flags]   -> /tmp/vittf_asm/<name>.s, resource usage on stdout
set -e
src=$1; shift
name=$(basename "$src" .hip)
mkdir -p /tmp/vittf_asm
extra=""
if [ "$name" = attention ]; then extra="-fno-honor-nans -mllvm -amdgpu-sched-strategy=iterative-ilp"; fi; if [ "$name" = attention_pipe ]; then extra="-fno-honor-nans"; fi
cd "$(dirname "$0")/../vit-tf_amd/csrc"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC $extra "$@" --cuda-device-only -S "$src" -o /tmp/vittf_asm/$name.s \
  -Rpass-analysis=kernel-resource-usage 2>&1 | grep -E "Function Name|VGPRs:|AGPRs|Spill|Occupancy|LDS Size|SGPRs:" | paste - - - - - - - - | sed 's/remark: [^ ]*//g' | cut -c1-400
