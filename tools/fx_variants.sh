#!/bin/bash
# Timing-only builds of the role-split block-tail kernel (csrc/tail_fx.hip, -DFX_VARIANT=bits: 1 main phase only, 2 no GELU
# arithmetic, 4 no LDS-DMA inside the steps, 8 no fragment refills, 16 in-kernel stamps), each as its own small shared object
# under tools/micro/build/ -- never part of libvittf.so.  Results of every variant other than 0 / 16 are wrong by construction.
#   tools/fx_variants.sh 17 19 21 "17:s4:-DFX_NSLOT=4" ...     then on the GPU box: python tools/fx_variants.py
set -e
tools=$(cd "$(dirname "$0")" && pwd)
cd "$tools/../vit-tf_amd/csrc"
line=$(make -n -B build/tail_fx.o | grep -- "-c tail_fx.hip" | head -1)
flags=$(echo "$line" | sed -e "s/ -c tail_fx.hip.*//" -e 's/^[^ ]*hipcc//')
mkdir -p "$tools/micro/build"
rm -f "$tools"/micro/build/libfx_v*.so
for spec in "$@"; do        # "bits" or "bits:tag:extra flag"
  v=${spec%%:*}; rest=${spec#*:}; tag=""; extra=""
  if [ "$rest" != "$spec" ]; then tag="_${rest%%:*}"; extra=${rest#*:}; fi
  /opt/rocm/bin/hipcc $flags -DFX_VARIANT=$v -DFX_STANDALONE $extra -shared tail_fx.hip -o "$tools/micro/build/libfx_v$v$tag.so" &
done
wait
ls -la "$tools/micro/build"
