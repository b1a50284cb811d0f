#!/bin/bash
# Timing-only builds of the whole-row GEMM (gemm_rows.hip: -DROWS_ABL_NO_MFMA / -DROWS_ABL_NO_EPI; wrong results) as
# separate libraries under tools/micro/build/, for tools/rows_ablate.py.  Run here (hipcc cross-compiles), then gpurun.
set -e
cd "$(dirname "$0")/../vit-tf_amd/csrc"
make >/dev/null
OUT=../../tools/micro/build
mkdir -p $OUT
OBJS=$(ls build/*.o | grep -v gemm_rows.o)
for v in full nomfma noepi neither; do
  case $v in
    full) D="" ;; nomfma) D="-DROWS_ABL_NO_MFMA" ;; noepi) D="-DROWS_ABL_NO_EPI" ;; neither) D="-DROWS_ABL_NO_MFMA -DROWS_ABL_NO_EPI" ;;
  esac
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC $D -c gemm_rows.hip -o $OUT/gemm_rows_$v.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $OUT/libvittf_rows_$v.so $OBJS $OUT/gemm_rows_$v.o
  rm -f $OUT/gemm_rows_$v.o
done
ls -la $OUT
