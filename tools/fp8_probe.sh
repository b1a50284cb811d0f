# GPU box: the fp8 attention path with q / k quantised in the qkv epilogue (parity tests, then ViT-B/8 end to end with / without it)
set -u
mkdir -p gpurun_out
timeout -k 10 400 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_pipeline.py -x -q -m gpu -s -p no:cacheprovider -k "fp8 or vitb8" > gpurun_out/fp8_tests.log 2>&1; echo "tests rc=$?"; grep -E "fp8|passed|failed|Error" gpurun_out/fp8_tests.log | tail -24
for e in "VITTF_FP8_ROWS=1" "VITTF_FP8_ROWS=0"; do
  export $e
  VITTF_BENCH_OVERLAP=0 VITTF_BENCH_E2E=0 timeout -k 10 400 python bench.py --arch vitb8 --attention fp8 --cpu-slices 0 --steps 2 --warmup 1 > gpurun_out/benchb512_fp8_$e.log 2>&1; echo "$e rc=$?"
  unset ${e%%=*}
done
