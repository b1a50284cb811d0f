for b in 32 8 12 16 32; do echo -n "engine batch $b: "; python bench.py --steps 2 --warmup 1 --cpu-slices 0 --workload 64 --engine-batch $b 2>&1 | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['roofline']['avg_launch_ms'], d['roofline']['kernel_ms_rank0'])"; done
