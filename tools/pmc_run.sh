#!/usr/bin/env bash
# usage: pmc_run.sh "<bench_kernels args>" "<counter set 1>" "<counter set 2>" ...   (one rocprofv3 --pmc pass per set)
set -u
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
OUT=gpurun_out/pmc
rm -rf $OUT; mkdir -p $OUT
WHAT=$1; shift
i=0
for set in "$@"; do
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
        k = row['Kernel_Name'][:70]
        agg[k][row['Counter_Name']].append(float(row['Counter_Value']))
for k, d in agg.items():
    if 'vectorized' in k or 'rocclr' in k or 'elementwise' in k: continue
    print(k)
    for c, v in sorted(d.items()):
        print(f'   {c:30s} mean {sum(v)/len(v):16.1f}  n={len(v)}')
PY
