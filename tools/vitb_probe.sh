# GPU box: the ViT-B/8 configuration end to end (BASELINE configs[3]), 16-bit and fp8 attention, and its parity tests
set -u
mkdir -p gpurun_out
timeout -k 10 400 python -m pytest tests/test_gpu_pipeline.py tests/test_gpu_kernels.py -x -q -m gpu -s -p no:cacheprovider -k "vitb8 or gemm or fp8 or kfeat" > gpurun_out/vitb_tests.log 2>&1; echo "tests rc=$?"; tail -8 gpurun_out/vitb_tests.log
for a in 16bit fp8; do
  VITTF_BENCH_OVERLAP=0 VITTF_BENCH_E2E=0 timeout -k 10 400 python bench.py --arch vitb8 --attention $a --cpu-slices 0 --steps 2 --warmup 1 > gpurun_out/benchb512_$a.log 2>&1; echo "$a rc=$?"
done
