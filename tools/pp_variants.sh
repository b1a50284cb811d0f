#!/bin/bash
# Timing-only builds of the ping-pong GEMM (csrc/gemm_pp.hip, -DPP_VARIANT=bits: 1 no LDS-DMA in the K loop, 2 DMA pieces
# without the M0 save / restore, 4 no fragment reads in the K loop, 8 two of a stage's four pieces issued at the head of the
# MFMA segment, 16 no s_setprio, 32 no epilogue), each as its own small shared object under tools/micro/build/ -- never part
# of libvittf.so.  Variants with bits 1, 4 or 32 compute wrong results by construction.
#   tools/pp_variants.sh 0 1 2 ...     then on the GPU box: python tools/pp_variants.py
set -e
tools=$(cd "$(dirname "$0")" && pwd)
cd "$tools/../vit-tf_amd/csrc"
line=$(make -n -B build/gemm_pp.o | grep -- "-c gemm_pp.hip" | head -1)
flags=$(echo "$line" | sed -e "s/ -c gemm_pp.hip.*//" -e 's/^[^ ]*hipcc//')
mkdir -p "$tools/micro/build"
rm -f "$tools"/micro/build/libpp_v*.so
for v in "$@"; do
  /opt/rocm/bin/hipcc $flags -DPP_VARIANT=$v -DPP_STANDALONE -shared gemm_pp.hip -o "$tools/micro/build/libpp_v$v.so" &
done
wait
ls -la "$tools/micro/build" | grep libpp
