set -u
mkdir -p gpurun_out
for e in "X=0" "VITTF_GEMM_ROWS=0" "VITTF_GEMM_STAGES=2"; do
  export $e
  timeout -k 10 400 python bench.py --arch vitb8 --cpu-slices 0 --steps 2 --warmup 1 > gpurun_out/vitb_$e.log 2>&1
  echo "$e rc=$?"
  unset ${e%%=*}
done
timeout -k 10 300 python bench.py --steps 2 --warmup 1 --cpu-slices 0 > gpurun_out/bench_sim.log 2>&1; echo rc=$?
