cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
export SETTLE_S=0.05
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/profsim -- python tools/bench_kernels.py sim > gpurun_out/profsim.log 2>&1
echo rc=$?
