#!/bin/bash
# Timing-only builds of the attention kernel (csrc/attention_pp64.hip, -DPP_DMA_EVERY=n: the LDS-DMA of every n-th key tile only,
# 0 = of the first three tiles only), each as its own small shared object under tools/micro/build/ -- never part of
# libvittf.so; results of every build other than 1 are wrong by construction.
#   tools/attn_variants.sh 1 2 0      then on the GPU box: python tools/attn_variants.py
set -e
tools=$(cd "$(dirname "$0")" && pwd)
cd "$tools/../vit-tf_amd/csrc"
line=$(make -n -B build/attention_pp64.o | grep -- "-c attention_pp64.hip" | head -1)
flags=$(echo "$line" | sed -e "s/ -c attention_pp64.hip.*//" -e 's/^[^ ]*hipcc//')
mkdir -p "$tools/micro/build"
rm -f "$tools"/micro/build/libattn_v*.so
for v in "$@"; do
  /opt/rocm/bin/hipcc $flags -DPP_DMA_EVERY=$v -DPP_STANDALONE -shared attention_pp64.hip -o "$tools/micro/build/libattn_v$v.so" &
done
wait
ls -la "$tools/micro/build" | grep libattn
