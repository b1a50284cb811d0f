#!/usr/bin/env bash
# Copies what a tools/gpu_round.sh pass left under gpurun_out/ (scratch, not tracked) into profiles/<tag>_* (tracked):
#   tools/collect_profiles.sh r03a
set -u
tag=$1
O=gpurun_out; P=profiles
last_json() { grep -E '^\{"metric"' "$1" | tail -1; }
[ -f $O/bench512.log ] && last_json $O/bench512.log > $P/${tag}_bench512.json
[ -f $O/rocprof512.log ] && last_json $O/rocprof512.log > $P/${tag}_bench512_under_rocprof.json
f=$(ls -t $O/prof512/runc/*kernel_stats.csv 2>/dev/null | head -1); [ -n "$f" ] && cp "$f" $P/${tag}_bench512_kernel_stats.csv
[ -f $O/pmc_attn256.log ] && cp $O/pmc_attn256.log $P/${tag}_attention_pmc.txt
[ -f $O/pmc_sim.log ] && cp $O/pmc_sim.log $P/${tag}_similarity_pmc.txt
[ -f $O/pmc_tail.log ] && cp $O/pmc_tail.log $P/${tag}_block_tail_pmc.txt
[ -f $O/pmc_attention.json ] && cp $O/pmc_attention.json $P/pmc_attention.json
[ -f $O/pmc_similarity.json ] && cp $O/pmc_similarity.json $P/pmc_similarity.json
[ -f $O/pmc_block_tail.json ] && cp $O/pmc_block_tail.json $P/pmc_block_tail.json
for t in kernels pipeline fullsize; do [ -f $O/test_$t.log ] && tail -n 40 $O/test_$t.log > $P/${tag}_test_$t.tail.txt; done
[ -f $O/test_all.log ] && tail -n 60 $O/test_all.log > $P/${tag}_test_all_one_process.tail.txt
[ -f $O/bench512_fos128.log ] && last_json $O/bench512_fos128.log > $P/${tag}_bench512_fos128.json
[ -f $O/smoke.log ] && tail -n 10 $O/smoke.log > $P/${tag}_smoke.tail.txt
for a in 16bit fp8; do [ -f $O/benchb512_$a.log ] && last_json $O/benchb512_$a.log > $P/${tag}_bench512_vitb8_$a.json; done
ls -la $P | grep "${tag}_" | awk '{print $5, $9}'
