#!/usr/bin/env bash
# Runs on the GPU box (via gpurun): GPU parity tests, smoke, a short bench and a rocprofv3 kernel trace.
# A step that times out or is killed stops the whole script (no further GPU work after a hang).
set -u
mkdir -p gpurun_out
OUT=gpurun_out
step() {  # step <name> <timeout-seconds> <cmd...>
  local name=$1 tmo=$2; shift 2
  echo "=== $name" | tee -a $OUT/summary.log
  timeout -k 10 "$tmo" "$@" > "$OUT/$name.log" 2>&1
  local rc=$?
  echo "rc=$rc" | tee -a $OUT/summary.log
  tail -n 25 "$OUT/$name.log"
  if [ $rc -ge 124 ]; then echo "step $name timed out / was killed: stopping" | tee -a $OUT/summary.log; exit $rc; fi
  return 0
}
: > $OUT/summary.log
for s in "$@"; do
  case $s in
    allgpu)   # what the driver runs at round end: the whole -m gpu suite as ONE process (900 s limit there)
              step test_all 900 python -m pytest tests -x -q -m gpu -s --durations=25 -p no:cacheprovider ;;
    fos128)   step bench512_fos128 900 python bench.py --fos 128 --steps 2 --warmup 1 --cpu-slices 0 ;;
    kernels)  step test_kernels 900 python -m pytest tests/test_gpu_kernels.py -q -m gpu -p no:cacheprovider ;;
    pipeline) step test_pipeline 900 python -m pytest tests/test_gpu_pipeline.py -q -m gpu -s -p no:cacheprovider ;;
    smoke)    step smoke 300 python -c "import __graft_entry__ as g; g.smoke()" ;;
    bench)    step bench 900 python bench.py --steps 2 --warmup 1 ;;
    bench64)  step bench64 600 python bench.py --steps 1 --warmup 1 --workload 64 --cpu-slices 0 ;;
    prof)     cd /tmp 2>/dev/null && export TMPDIR=/tmp && cd - >/dev/null
              step rocprof 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -- python bench.py --steps 1 --warmup 1 --workload 64 --cpu-slices 0 ;;
    attn)     step test_attn 900 python -m pytest tests/test_gpu_kernels.py -q -m gpu -p no:cacheprovider -k "attention or resize or widened or empty_class or single_annotation" ;;
    fullsize) step test_fullsize 1100 python -m pytest tests/test_gpu_fullsize.py -q -m gpu -s -p no:cacheprovider ;;
    ceiling)  step ceiling 240 tools/micro/mfma_ceiling ;;
    pmcattn)  step pmc_attn 900 bash tools/pmc_attn.sh attn ;;
    pmcsim)   step pmc_sim 600 bash tools/pmc_sim.sh ;;
    pmcjson)  # the records bench.py reads for roofline.traffic: attention at the engine's batch (256 slices), the 16-query similarity
              export BATCH=512 DT=fp16
              step pmc_attn256 900 bash tools/pmc_attn.sh attn
              unset BATCH
              python tools/pmc_json.py attention $OUT/pmc_attn256.log attn_pp64 $OUT/pmc_attention.json batch=512 tokens=4097 heads=6 "kernel=attn_pp64_kernel<fp16>" > /dev/null
              step pmc_sim 600 bash tools/pmc_sim.sh
              python tools/pmc_json.py similarity $OUT/pmc_sim.log sim_mfma_few $OUT/pmc_similarity.json batch=16 nvox=262144 features=384 kernel=sim_mfma_few_kernel > /dev/null ;;
    pmctail)  export BATCH=512 DT=fp16
              step pmc_tail 900 bash tools/pmc_attn.sh tail
              unset BATCH
              python tools/pmc_json.py block_tail $OUT/pmc_tail.log tail_fx_kernel $OUT/pmc_block_tail.json batch=512 tokens=4097 features=384 > /dev/null ;;
    pmcqkv)   export BATCH=512 DT=fp16
              step pmc_qkv 900 bash tools/pmc_attn.sh qkv
              unset BATCH
              python tools/pmc_json.py gemm_qkv $OUT/pmc_qkv.log gemm_as_kernel $OUT/pmc_gemm_qkv.json batch=512 tokens=4097 features=384 > /dev/null ;;
    fp8)      step test_fp8 600 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_pipeline.py -q -m gpu -s -p no:cacheprovider -k "fp8 or vitb8" ;;
    benchb)   for a in 16bit fp8; do step benchb_$a 600 python bench.py --arch vitb8 --workload 64 --attention $a --cpu-slices 0 --steps 2; done ;;
    benchb512) for a in 16bit fp8; do step benchb512_$a 600 python bench.py --arch vitb8 --attention $a --cpu-slices 0 --steps 2; done ;;
    simmany)  step sim_many 300 python tools/sim_many_profile.py ;;
    simtests) step test_sim 600 python -m pytest tests/test_gpu_kernels.py -q -m gpu -p no:cacheprovider -k "similarity or sim or golden or labels or cosine or topk" ;;
    prof8)    cd /tmp 2>/dev/null && export TMPDIR=/tmp && cd - >/dev/null
              SETTLE_S=0.05 step prof8 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof8 -- python tools/bench_kernels.py attn8 ;;
    benchbatch) for eb in ${BENCH_BATCHES:-64 128 256 512}; do step benchbatch_$eb 600 python bench.py --engine-batch $eb --cpu-slices 0 --steps 2; done ;;
    bench512) step bench512 900 python bench.py ;;
    prof512)  cd /tmp 2>/dev/null && export TMPDIR=/tmp && cd - >/dev/null
              export VITTF_BENCH_EXTRAS=0
              step rocprof512 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof512 -- python bench.py --steps 1 --warmup 1 --cpu-slices 0 ;;
    *) echo "unknown step $s" ;;
  esac
done
echo "=== done" | tee -a $OUT/summary.log
