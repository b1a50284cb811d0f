#!/usr/bin/env bash
# PMC passes over the attention microbench (counters only with --kernel-trace; separate runs per counter set).
set -u
export SETTLE_S=0.05   # (bench_kernels.py: no clock-settling loop under the profiler)
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
OUT=gpurun_out/pmc
rm -rf $OUT; mkdir -p $OUT
WHAT=${1:-attn}
i=0
for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" \
           "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_SALU" \
           "GRBM_GUI_ACTIVE TCC_HIT_sum TCC_MISS_sum" \
           "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $OUT/p$i -- python tools/bench_kernels.py $WHAT > $OUT/p$i.log 2>&1
  rc=$?
  echo "pass $i rc=$rc"
  if [ $rc -ge 124 ]; then exit $rc; fi
done
python - <<'PY'
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob('gpurun_out/pmc/p*/**/*counter_collection.csv', recursive=True):
    for row in csv.DictReader(open(f)):
        k = row['Kernel_Name'][:60]
        agg[k][row['Counter_Name']].append(float(row['Counter_Value']))
for k, d in agg.items():
    if 'vectorized' in k or 'rocclr' in k: continue
    print(k)
    for c, v in sorted(d.items()):
        print(f'   {c:28s} mean {sum(v)/len(v):16.1f}  n={len(v)}')
PY
