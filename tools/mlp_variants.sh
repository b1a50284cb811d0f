#!/bin/bash
# Timing-only builds of the fused MLP kernel (csrc/mlp.hip, -DMLP_VARIANT=bits: 1 no weight DMA inside the units, 2 LDS-DMA
# WITH the M0 save / restore, 4 no fragment refills, 8 no activation arithmetic, 16 in-kernel stamps, 32 / 64 block tail:
# projection units without the residual chunk requests / without folding them in), each as its own small shared object
# under tools/micro/build/ -- never part of libvittf.so.  Results of variants other than 0 and 2 are wrong by construction.
#   tools/mlp_variants.sh 0 1 2 4 8 ...     then on the GPU box: python tools/mlp_variants.py
set -e
tools=$(cd "$(dirname "$0")" && pwd)
cd "$tools/../vit-tf_amd/csrc"
line=$(make -n -B build/mlp.o | grep -- "-c mlp.hip" | head -1)
flags=$(echo "$line" | sed -e "s/ -c mlp.hip.*//" -e 's/^[^ ]*hipcc//')
mkdir -p "$tools/micro/build"
rm -f "$tools"/micro/build/libmlp_v*.so
for spec in "$@"; do        # "bits" or "bits:tag:extra flag" (e.g. 0:1:-DMLP_HP=1 -> libmlp_v0_1.so)
  v=${spec%%:*}; rest=${spec#*:}; tag=""; extra=""
  if [ "$rest" != "$spec" ]; then tag="_${rest%%:*}"; extra=${rest#*:}; fi
  /opt/rocm/bin/hipcc $flags -DMLP_VARIANT=$v -DMLP_STANDALONE $extra -shared mlp.hip -o "$tools/micro/build/libmlp_v$v$tag.so" &
done
wait
ls -la "$tools/micro/build"
