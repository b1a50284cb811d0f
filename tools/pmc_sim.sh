#!/usr/bin/env bash
# PMC passes (HBM bytes) over the similarity microbench: FETCH_SIZE / WRITE_SIZE in separate runs, --kernel-trace only.
set -u
export SETTLE_S=0.05   # (bench_kernels.py: no clock-settling loop under the profiler)
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
OUT=gpurun_out/pmc_sim
rm -rf $OUT; mkdir -p $OUT
i=0
for set in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU"; do
  i=$((i+1))
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $OUT/p$i -- python tools/bench_kernels.py sim > $OUT/p$i.log 2>&1
  rc=$?
  echo "pass $i rc=$rc"
  if [ $rc -ge 124 ]; then exit $rc; fi
done
python - <<'PY'
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob('gpurun_out/pmc_sim/p*/**/*counter_collection.csv', recursive=True):
    for row in csv.DictReader(open(f)):
        k = row['Kernel_Name'][:70]
        agg[k][row['Counter_Name']].append(float(row['Counter_Value']))
for k, d in agg.items():
    if 'sim_' not in k: continue
    print(k)
    for c, v in sorted(d.items()):
        print(f'   {c:28s} mean {sum(v)/len(v):16.1f}  n={len(v)}')
PY
