# GPU box: parity tests of the GEMM class, then the ViT-B linears (D = 768) with / without the ping-pong kernel
set -u
mkdir -p gpurun_out
timeout -k 10 300 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -p no:cacheprovider -k "gemm" > gpurun_out/pp_tests.log 2>&1; echo "tests rc=$?"; tail -5 gpurun_out/pp_tests.log
for e in "VITTF_GEMM_PP=1" "VITTF_GEMM_PP=0" "VITTF_PP_RESIDUAL=1"; do
  export $e
  D=768 BATCH=${BATCH:-64} timeout -k 10 200 python tools/bench_kernels.py gemm > gpurun_out/pp_bench_$e.log 2>&1; echo "$e rc=$?"; grep gemm gpurun_out/pp_bench_$e.log
  unset ${e%%=*}
done
