#!/bin/bash
# Device assembly + register / occupancy report of one kernel source, built with EXACTLY the flags the Makefile uses for it
# (taken from `make -n`, so the two cannot drift apart).
#   tools/kernel_asm.sh attention_pp64.hip [extra flags]   -> /tmp/vittf_asm/<name>.s, resource usage on stdout,
#   then tools/asm_hot_scratch.py on the result (spill instructions inside pinned loop blocks)
set -e
tools=$(cd "$(dirname "$0")" && pwd)
src=$1; shift
name=$(basename "$src" .hip)
mkdir -p /tmp/vittf_asm
cd "$tools/../vit-tf_amd/csrc"
# the compile line make would run for this object (-B: even if it is up to date), minus its "-c src -o obj" tail
line=$(make -n -B build/$name.o | grep -- "-c $name.hip" | head -1)
flags=$(echo "$line" | sed -e "s/ -c $name.hip.*//" -e 's/^[^ ]*hipcc//')
/opt/rocm/bin/hipcc $flags "$@" --cuda-device-only -S "$name.hip" -o /tmp/vittf_asm/$name.s \
  -Rpass-analysis=kernel-resource-usage 2>&1 | grep -E "Function Name|VGPRs:|AGPRs|Spill|ScratchSize|Occupancy|LDS Size|SGPRs:" | paste - - - - - - - - - | sed 's/remark: [^ ]*//g; s/\[-Rpass-analysis=kernel-resource-usage\]//g' | sed "s/$name.hip:[0-9]*:[0-9]*: //g" | tr -s ' \t' ' ' | cut -c1-300
python3 "$tools/asm_hot_scratch.py" /tmp/vittf_asm/$name.s
