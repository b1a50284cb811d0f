#!/bin/bash
# Timing-only builds of the persistent 256 x 256 GEMM (csrc/gemm_pp.hip, -DPP_VARIANT=bits: 1 no LDS-DMA in the K loop, 4 no
# fragment reads in the K loop, 32 no epilogue, 128 no scheduling fences inside a stage, 512 / 1024 / 2048 the epilogue's
# 16-bit stores with the plain / sc0 nt / sc0 sc1 cache policy instead of nt), each as its own small shared object under
# tools/micro/build/ -- never part of libvittf.so.  Variants with bits 1, 4 or 32 compute wrong results by construction.
# (Bits 2, 8, 16 and 64 belonged to the two-segment forms of the kernel: LAB_NOTES.md, former DESIGN section 4.)
#   tools/pp_variants.sh 0 1 2 ...     then on the GPU box: python tools/pp_variants.py
set -e
tools=$(cd "$(dirname "$0")" && pwd)
cd "$tools/../vit-tf_amd/csrc"
mkdir -p "$tools/micro/build"
rm -f "$tools"/micro/build/libpp_v*.so
for spec in "$@"; do
  name=gemm_pp; v=$spec; tag=pp
  line=$(make -n -B build/$name.o | grep -- "-c $name.hip" | head -1)
  flags=$(echo "$line" | sed -e "s/ -c $name.hip.*//" -e 's/^[^ ]*hipcc//')
  /opt/rocm/bin/hipcc $flags -DPP_VARIANT=$v -DPP_STANDALONE -shared $name.hip -o "$tools/micro/build/lib${tag}_v$v.so" &
done
wait
ls -la "$tools/micro/build" | grep libpp
