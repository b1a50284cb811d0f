#!/usr/bin/env python3
"""Benchmark of the vit-tf hot path on MI355X: slices/sec of the ViT-S/8 feature-volume extraction on a 512^3 volume
(+ the 16-query similarity step), BASELINE.json's metric.

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

`python bench.py --gpus N` with N > 1 and no launcher around it starts the N ranks itself (child processes under
torch.distributed.run, before this process touches a GPU) and exits with their code.

One "step" = one pass of the hot path over one synthetic volume that is already resident in HBM: every axis-aligned slice
through ViT-S/8 (seeded synthetic weights, fp16 MFMA operands = the reference's GPU autocast type, the dtype whose parity
tests assert 1e-3), slice-axis pooling, the fp16 z+y+x sum (with one RCCL all-gather per axis when N > 1), then a 16-query
similarity volume and the label volume.  Every N runs the metric's configuration: the 512^3 CT-like volume = 1536 slices
of 512 x 512 per step (BASELINE configs[2]; sharded over the ranks for N > 1, "scaling": "strong").  value = slices pushed
through the ViT by all ranks / wall time of the K timed steps (max over ranks).

Extra objects on the JSON line: "roofline" for the kernel with the largest share of the step (durations from HIP events
recorded on the launch stream inside the timed region, algorithmic FLOPs from SURVEY.md 8d), "roofline_similarity" for
the similarity accumulation kernel (HBM-bound), and "cpu_baseline" (the oracle's CPU fp32 restatement of the same ViT and
of the reference's similarity formulation on the host cores, rank 0 at N = 1, bounded samples).
"""
import argparse
import hashlib
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

PEAK_TFLOPS = {'bf16': 2500.0, 'fp16': 2500.0, 'fp8': 5000.0}     # dense MFMA peaks, MI355X_MICROARCH.md
PEAK_HBM_GBS = 8000.0                              # HBM3E peak, MI355X_MICROARCH.md (6.3 TB/s achievable)
FOS = 64
N_QUERIES = 16


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=3)
    ap.add_argument('--warmup', type=int, default=1)
    ap.add_argument('--workload', type=str, default='512', choices=['64', '256', '512'],
                    help="512 = the metric's configuration (default for every N); 256 / 64 = BASELINE configs[1] / [0]")
    ap.add_argument('--dtype', type=str, default='fp16', choices=['fp16', 'bf16'],
                    help='MFMA operand type: fp16 (default; parity tests assert 1e-3) or bf16 (2.4e-3 .. 3.9e-3)')
    ap.add_argument('--arch', type=str, default='vits8', choices=['vits8', 'vitb8'],
                    help="vits8 = the BASELINE metric's model; vitb8 = configs[3]'s feature extractor (D = 768), informative")
    ap.add_argument('--attention', type=str, default='16bit', choices=['16bit', 'fp8'],
                    help="fp8 = BASELINE configs[3]'s fp8 MFMA attention path (informative: ~3e-2 on the features, not the contract dtype)")
    ap.add_argument('--fos', type=int, default=FOS, help='--feature-output-size (64: the metric; 128: the sub/infer_and_merge.sh preset, 1024 x 1024 slices, N = 16385)')
    ap.add_argument('--engine-batch', type=int, default=None, help='slices per engine call (default: extract.DEFAULT_ENGINE_BATCH)')
    ap.add_argument('--cpu-slices', type=int, default=5, help='slices timed for the CPU baseline (0 = skip)')
    return ap.parse_args()


def spawn_ranks(args):
    """No launcher around us and --gpus N > 1: start the N ranks as children (never a re-exec) and leave with their code."""
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', f'--nproc-per-node={args.gpus}',
           '--master-addr', '127.0.0.1', '--master-port', str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    return subprocess.call(cmd, env=env)


def vit_flops(n_tokens, dim, depth, patch):
    """Algorithmic FLOPs per slice: per-kernel-class shares of F_min."""
    nd2 = n_tokens * dim * dim
    return {
        'attention': (depth - 1) * 4 * n_tokens * n_tokens * dim,
        'gemm_qkv': (depth - 1) * 6 * nd2, 'gemm_proj': (depth - 1) * 2 * nd2,
        'gemm_fc1': (depth - 1) * 8 * nd2, 'gemm_fc2': (depth - 1) * 8 * nd2,
        'gemm': 2 * nd2,                                     # K projection of the last block
        'patch_embed': 2 * (n_tokens - 1) * 3 * patch * patch * dim,
    }


def extra_configs(vt, torch, dvol, vol, feats, dev, dtype):
    """BASELINE configs[3] (ViT-B/8, 16-bit and fp8 attention), configs[4] (5 x 1024-query similarity + the bilateral solver) and the
    fos-128 preset of sub/infer_and_merge.sh on the SAME 512^3 volume, behind the contract's timed region: a short run each,
    with the dominant kernel class priced against its own peak.  Informative objects on the JSON line, never `value`."""
    out = {}
    total_slices = sum(dvol.shape)

    def extractor(arch, attention, fos, steps):
        dim, depth, heads, patch = vt.ARCHS[arch]
        model = vt.HipViT(vt.synthetic_state_dict(arch, 0), arch, dtype, device=dev, attention=attention)
        im, fo = vt.sizing(dvol.shape, fos, 8)
        ntok = (im[0] // 8) * (im[1] // 8) + 1
        eb = vt.extract.engine_batch_for(ntok, dim)
        run = lambda: vt.feature_volume(None, model, fos, 'all', eb, dvol=dvol)
        run()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            run()
        torch.cuda.synchronize()
        el = (time.perf_counter() - t0) / steps
        vt._lib.profiler_enable(True)
        run()
        torch.cuda.synchronize()
        prof = vt._lib.profiler_collect()
        vt._lib.profiler_enable(False)
        fl = vit_flops(ntok, dim, depth, patch)
        if prof['mlp'][1] > 0:
            fl['mlp'] = fl.pop('gemm_fc1') + fl.pop('gemm_fc2') + fl.pop('gemm_proj')
        fl = {k: v * total_slices for k, v in fl.items()}
        tf = {k: round(fl[k] / (prof[k][0] * 1e-3) / 1e12, 1) for k in fl if prof[k][0] > 0}
        dom = max(tf, key=lambda k: prof[k][0])
        peak = PEAK_TFLOPS['fp8' if (dom == 'attention' and attention == 'fp8') else dtype]
        del model
        torch.cuda.empty_cache()
        return {'value': round(total_slices / el, 1), 'unit': 'slices/s', 'ms_per_step': round(el * 1e3, 1), 'steps': steps, 'tokens': ntok,
                'engine_batch': eb, 'attention': attention, 'dominant_kernel': dom, 'dominant_share': round(prof[dom][0] / sum(v[0] for v in prof.values()), 3),
                'frac': round(tf[dom] / peak, 4), 'peak_tflops': peak, 'kernel_tflops': tf,
                'whole_vit_tflops': round(sum(fl.values()) / (sum(v[0] for k, v in prof.items() if k != 'similarity') * 1e-3) / 1e12, 1)}
    out['config3_vitb8'] = {'16bit': extractor('vitb8', '16bit', FOS, 2), 'fp8': extractor('vitb8', 'fp8', FOS, 2),
                            'note': 'BASELINE configs[3] on one GPU: ViT-B/8 (D = 768) feature volume of the same 512^3 volume'}
    out['fos128'] = dict(extractor('vits8', '16bit', 128, 1), note='sub/infer_and_merge.sh preset: 1024 x 1024 slices, N = 16385')
    # configs[4]: five classes of 1024 annotations each on the metric's feature volume (seeded voxel coordinates)
    g = torch.Generator().manual_seed(4)
    ann = {f'c{i}': torch.stack([torch.randint(0, n, (1024,), generator=g) for n in dvol.shape], dim=1) for i in range(5)}
    vt.compute_similarities(vol, feats, ann, keep_on_device=True)
    torch.cuda.synchronize()
    vt._lib.profiler_enable(True, classes=['similarity'])
    t0 = time.perf_counter()
    for _ in range(5):
        vt.compute_similarities(vol, feats, ann, keep_on_device=True)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / 5 * 1e3
    k_ms, k_n = vt._lib.profiler_collect()['similarity']
    vt._lib.profiler_enable(False)
    nvox, dim = feats[0].numel(), feats.shape[0]
    fl = 2.0 * nvox * dim * 5120 * 2            # queries enter as fp16 hi + lo halves: two matrix products
    t0 = time.perf_counter()
    vt.compute_similarities(vol, feats, ann, keep_on_device=True, bilateral_solver=True)
    torch.cuda.synchronize()
    bls_ms = (time.perf_counter() - t0) * 1e3
    out['config4_1024q'] = {'queries': 5120, 'ms': round(ms, 3), 'mvoxel_sim_per_s': round(nvox * 5120 / 1e6 / (ms * 1e-3), 1),
                            'kernel': vt._lib.kernel_name('similarity'), 'kernel_ms': round(k_ms / max(1, k_n), 3),
                            'kernel_tflops': round(fl / (k_ms / max(1, k_n) * 1e-3) / 1e12, 1), 'frac': round(fl / (k_ms / max(1, k_n) * 1e-3) / 1e12 / PEAK_TFLOPS['fp16'], 4),
                            'with_bilateral_solver_ms': round(bls_ms, 2),
                            'note': 'BASELINE configs[4]: 5 x 1024 annotations on the (384, 64, 64, 64) volume, maps left on the GPU; frac: hi + lo '
                                    'query halves = two matrix products, against the fp16 MFMA peak; the bilateral-solver call includes its first-call setup'}
    return out


def make_workload(name, vt):
    """(volume fp16, label uint8, description) -- seeded, generated on the host before timing."""
    if name == '256':
        vol, label = vt.synthetic_volume('torus_filled', 256, 0.1, 0)
        return vol, label, '256^3 synthetic torus volume (noise 0.1, seed 0), ViT-S/8, fos 64: 768 slices of 512x512 ' \
                           '(N=4097) + 16-query similarity -> BASELINE configs[1]'
    if name == '512':
        vol, label = vt.ct_like_volume(512, 0)
        return vol, label, '512^3 CT-like synthetic volume (seed 0), ViT-S/8, fos 64: 1536 slices of 512x512 (N=4097) per ' \
                           'step + 16-query similarity -> the metric\'s configuration (BASELINE configs[2])'
    if name == '64':
        vol, label = vt.synthetic_volume('torus_filled', 64, 0.1, 0)
        return vol, label, '64^3 synthetic torus volume, ViT-S/8, fos 64: 192 slices of 512x512 (N=4097) -> BASELINE configs[0]'
    raise SystemExit(f'unknown workload {name}')


def query_voxels(label, n=N_QUERIES):
    """n seeded voxel coordinates inside the labelled region (drawn from a strided sub-grid: cheap at 512^3)."""
    import torch
    g = torch.Generator().manual_seed(0)
    step = max(1, label.shape[0] // 64)
    idx = (label[::step, ::step, ::step] > 0).nonzero() * step
    pick = torch.randperm(idx.shape[0], generator=g)[:n]
    return {'ntf1': idx[pick]}


KERNEL_SOURCES = {     # what a kernel's code object is built from: its source, the shared headers and the Makefile's flags
    'attention': ('attention_pp64.hip', 'attn_common.h', 'vittf_common.h', 'Makefile'),
    'similarity': ('sim_mfma.hip', 'similarity.hip', 'vittf_common.h', 'Makefile'),
    'block_tail': ('tail_fx.hip', 'vittf_common.h', 'Makefile'),
    'gemm_qkv': ('gemm_as.hip', 'vittf_common.h', 'Makefile'),
}


def kernel_source_hash(kernel_class):
    """sha1 over everything the kernel of a class is compiled from: a committed PMC pass only describes the code object it
    was measured on (a header or a flag change can add a spill, and with it HBM traffic)."""
    h = hashlib.sha1()
    for name in KERNEL_SOURCES[kernel_class]:
        with open(os.path.join(ROOT, 'vit-tf_amd', 'csrc', name), 'rb') as f:
            h.update(name.encode() + b'\0' + f.read())
    return h.hexdigest()


def pmc_traffic(kernel_class, batch, **shape):
    """HBM bytes per launch of a kernel from the committed rocprofv3 PMC pass (profiles/pmc_*.json: FETCH_SIZE doubled
    as MI355X_MICROARCH.md prescribes for gfx950 wide reads + WRITE_SIZE, separate passes).  The counters cannot be
    collected from inside this process; null when no pass is on file for this very code object, shape AND batch (the
    pass is taken at the batch the engine runs: tools/pmc_attn.sh, tools/pmc_json.py)."""
    path = os.path.join(ROOT, 'profiles', f'pmc_{kernel_class}.json')
    try:
        rec = json.load(open(path))
    except (OSError, ValueError):
        return None
    if rec.get('batch') != batch or rec.get('source_sha1') != kernel_source_hash(kernel_class):
        return None
    if any(rec.get(k) != v for k, v in shape.items()):      # a pass taken on another shape (heads, feature width) says nothing here
        return None
    return int(rec['hbm_bytes_per_launch'])


def host_cores():
    """Cores this process may actually use: affinity mask capped by the cgroup CPU quota (a GPU box exposes all
    256 host CPUs to os.cpu_count() but grants a 16-CPU share; oversubscribing it makes torch 10x slower)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, 'sched_getaffinity') else (os.cpu_count() or 1)
    try:
        quota, period = open('/sys/fs/cgroup/cpu.max').read().split()
        if quota != 'max':
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return max(1, min(n, int(os.environ.get('VITTF_CPU_THREADS', '16'))))


def cpu_baseline(sd, vol, n_slices, im_sz, arch, feats_cpu, ann, vol_shape):
    """The oracle (CPU restatement of the reference path) on the host cores: batch-1 slice loop in fp32, and the
    reference's similarity formulation (einsum + where / pow / mean + quantise, predict_ntf.py:65, 71-72, 98-100)."""
    import torch
    from oracle import dino_vit, feature_volume as ofv, similarity as osim
    cores = host_cores()
    torch.set_num_threads(cores)
    model = dino_vit.build_vit(arch, sd)
    # the z-slices around the middle of the volume, normalised exactly as the whole volume would be (global min / max)
    mid = vol.shape[2] // 2
    sub = vol[:, :, mid:mid + n_slices + 1].float()
    lo, hi = float(vol.float().min()), float(vol.float().max())
    imgs = ofv.normalized_slices(sub, 'z', minmax=(lo, hi))
    times = []
    with torch.no_grad():
        for i in range(imgs.shape[0]):
            t0 = time.perf_counter()
            x = torch.nn.functional.interpolate(imgs[i:i + 1], size=(im_sz[0], im_sz[1]), mode='nearest')
            ofv.k_tokens(model, x).half()
            if i > 0:
                times.append(time.perf_counter() - t0)
    out = {'value': round(len(times) / sum(times), 4), 'unit': 'slices/s', 'cores': cores, 'kind': 'port',
           'sample': f'{len(times)} z-slices of the same volume at {im_sz[0]}x{im_sz[1]} (N={(im_sz[0] // 8) * (im_sz[1] // 8) + 1}), batch 1, fp32 torch CPU, '
                     f'after 1 warm-up slice; the oracle runs the K projection of block 12 only, like the GPU path'}
    # similarity: the full 64^3 x 384 feature volume, A = 16, the reference's own formulation
    f32 = feats_cpu.float()
    osim.similarity_maps(vol_shape, f32, ann)                       # warm-up
    reps, t0 = 3, time.perf_counter()
    for _ in range(reps):
        osim.similarity_maps(vol_shape, f32, ann)
    dt = (time.perf_counter() - t0) / reps
    nvox = f32[0].numel()
    out['similarity'] = {'ms': round(dt * 1e3, 2), 'mvoxel_sim_per_s': round(nvox * N_QUERIES / 1e6 / dt, 1), 'queries': N_QUERIES,
                         'cores': cores, 'kind': 'port',
                         'sample': f'full {tuple(f32.shape)} fp32 feature volume, {N_QUERIES} queries, einsum + where/pow/mean + '
                                   f'quantise + nearest resize (oracle/similarity.py = predict_ntf.py:52-72, 95-100), mean of {reps}'}
    return out


def main():
    args = parse_args()
    if 'WORLD_SIZE' not in os.environ and args.gpus > 1:
        sys.exit(spawn_ranks(args))                       # before anything touches a GPU

    import torch
    import vit_tf_amd as vt

    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local = int(os.environ.get('LOCAL_RANK', '0'))
    if world != args.gpus:
        raise SystemExit(f'--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {args.gpus}')
    local = local % max(1, torch.cuda.device_count())     # (rehearsals put several ranks on one GPU)
    torch.cuda.set_device(local)
    dev = torch.device('cuda', local)
    backend = None
    if world > 1:
        backend = os.environ.get('VITTF_DIST_BACKEND', 'nccl')      # 'nccl' is RCCL on ROCm
        if backend == 'nccl':
            torch.distributed.init_process_group('nccl', device_id=dev)
        else:
            torch.distributed.init_process_group(backend)
    barrier = (lambda: torch.distributed.barrier()) if world > 1 else (lambda: None)

    def max_over_ranks(x):
        if world == 1:
            return x
        t = torch.tensor([x], dtype=torch.float64, device=dev if backend == 'nccl' else 'cpu')
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        return float(t.item())

    torch.set_num_threads(max(1, host_cores() // max(1, min(world, 8))))   # host-side generation only
    vol, label, desc = make_workload(args.workload, vt)
    desc = desc.replace('ViT-S/8', 'ViT-B/8 (D = 768)') if args.arch == 'vitb8' else desc
    if args.fos != FOS:
        desc = desc.replace(f'fos {FOS}', f'fos {args.fos}').replace('512x512 (N=4097)', f'{8 * args.fos}x{8 * args.fos} (N={args.fos * args.fos + 1})') + ' [non-default --fos: not the metric\'s configuration]'
    sd = vt.synthetic_state_dict(args.arch, 0)
    model = vt.HipViT(sd, args.arch, args.dtype, device=dev, attention=args.attention)
    dim, depth, heads, patch = vt.ARCHS[args.arch]
    dvol = vt.DeviceVolume(vol, dev)                     # the input is resident in HBM before the timed region
    ann = query_voxels(label)
    im_sz, feat_out = vt.sizing(dvol.shape, args.fos, 8)
    n_tokens = (im_sz[0] // 8) * (im_sz[1] // 8) + 1
    args.engine_batch = vt.extract.engine_batch_for(n_tokens, vt.ARCHS[args.arch][0], args.engine_batch)
    total_slices = sum(dvol.shape)
    my_slices = 0
    for sl in range(3):
        w0, nw, _ = vt.extract.shard_windows(feat_out[sl], rank, world)
        if nw:
            my_slices += vt.extract.window_bounds(w0 + nw - 1, dvol.shape[sl], feat_out[sl])[1] - \
                vt.extract.window_bounds(w0, dvol.shape[sl], feat_out[sl])[0]

    def step(dvol=dvol):
        feats = vt.feature_volume(None, model, args.fos, 'all', args.engine_batch, dvol=dvol)
        sims = vt.compute_similarities(vol, feats, ann, keep_on_device=True)     # maps feed the label kernel directly
        return feats, vt.assign_labels(sims)

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    barrier()
    # HIP events around the launches of the dominant kernel only (attention, ~50 % of the kernel time): bracketing every
    # launch costs 2-3 % of the throughput.  The other classes are timed in one extra, untimed step below.
    fullprof = os.environ.get('VITTF_BENCH_FULLPROF') == '1'
    vt._lib.profiler_enable(True, classes=None if fullprof else ['attention'])
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        feats, labels = step()
    torch.cuda.synchronize()
    barrier()
    elapsed = time.perf_counter() - t0
    prof = vt._lib.profiler_collect()
    if not fullprof:
        vt._lib.profiler_enable(True)                       # all classes, one step outside the timed region
        step()
        torch.cuda.synchronize()
        prof_all = vt._lib.profiler_collect()
        for k, (ms, n) in prof_all.items():
            if k != 'attention':
                prof[k] = (ms * args.steps, n * args.steps)  # scaled to the timed steps (same launches every step)
    vt._lib.profiler_enable(False)
    elapsed = max_over_ranks(elapsed)

    # the reference's own bracket (infer.py:324 -> 336) also contains what bench.py's contract keeps OUT of the timed region:
    # the volume's way to the device (host fp16 -> HBM, infer.py:177 `.to(dev)`), `vol.float()` (:137) and the global
    # min / max (:155).  Measured here for the same K steps, behind the timed region: every step starts from the HOST volume.
    e2e = None
    if os.environ.get('VITTF_BENCH_E2E', '1') == '1':
        up = []
        torch.cuda.synchronize()
        barrier()
        t1 = time.perf_counter()
        for _ in range(args.steps):
            tu = time.perf_counter()
            dv = vt.DeviceVolume(vol, dev)               # H2D of the 2-byte volume + widening + finiteness check + min / max
            torch.cuda.synchronize()
            up.append(time.perf_counter() - tu)
            step(dv)
            del dv
        torch.cuda.synchronize()
        barrier()
        el3 = max_over_ranks(time.perf_counter() - t1)
        e2e = {'value': round(args.steps * total_slices / el3, 2), 'unit': 'slices/s', 'ms_per_step': round(el3 / args.steps * 1e3, 2),
               'upload_ms': round(sum(up) / len(up) * 1e3, 2), 'volume_bytes_host': int(vol.numel() * vol.element_size()),
               'note': 'every step starts from the host volume: H2D + widen to fp32 + finiteness check + global min / max inside '
                       'the bracket, as in infer.py:324-336 (pageable host memory, PCIe-inclusive; never the contract value)'}

    # similarity leg alone (outside the timed region): Mvoxel-sim/s = Nvox * A / time; HIP events around the accumulation
    # kernel (class 'similarity') give its HBM roofline
    n_rep = 20
    vt.compute_similarities(vol, feats, ann)
    torch.cuda.synchronize()
    ts = time.perf_counter()
    for _ in range(n_rep):
        vt.compute_similarities(vol, feats, ann)
    torch.cuda.synchronize()
    sim_ms = (time.perf_counter() - ts) / n_rep * 1e3
    vt._lib.profiler_enable(True, classes=['similarity'])
    ts = time.perf_counter()
    for _ in range(n_rep):                               # the same query with the maps left on the GPU (no host copies)
        vt.compute_similarities(vol, feats, ann, keep_on_device=True)
    torch.cuda.synchronize()
    sim_dev_ms = (time.perf_counter() - ts) / n_rep * 1e3
    sim_k_ms, sim_k_n = vt._lib.profiler_collect()['similarity']
    vt._lib.profiler_enable(False)
    nvox = feat_out[0] * feat_out[1] * feat_out[2]
    n_classes = len(ann)
    sim_bytes = nvox * (2 * dim + n_classes)                 # the fp16 feature row read once + one map byte per class
    sim_avg_ms = sim_k_ms / max(1, sim_k_n)
    sim_gbs = sim_bytes / (sim_avg_ms * 1e-3) / 1e9 if sim_avg_ms > 0 else 0.0
    sim_kernel = vt._lib.kernel_name('similarity')           # what the library's dispatcher launched, not a re-derivation
    # the same query over three copies of the volume in turn (3 x 201 MB > the 256 MB Infinity Cache): every launch
    # streams its volume from HBM -- the cold figure next to the cache-resident one above
    copies = [feats] + [feats.clone() for _ in range(2)]
    for c in copies:
        vt.compute_similarities(vol, c, ann, keep_on_device=True)
    torch.cuda.synchronize()
    vt._lib.profiler_enable(True, classes=['similarity'])
    for i in range(3 * 7):
        vt.compute_similarities(vol, copies[i % 3], ann, keep_on_device=True)
    torch.cuda.synchronize()
    cold_ms, cold_n = vt._lib.profiler_collect()['similarity']
    vt._lib.profiler_enable(False)
    del copies
    cold_avg_ms = cold_ms / max(1, cold_n)
    cold_gbs = sim_bytes / (cold_avg_ms * 1e-3) / 1e9 if cold_avg_ms > 0 else 0.0
    roofline_sim = {
        'bound': 'hbm', 'achieved': round(sim_gbs, 1), 'peak': PEAK_HBM_GBS, 'unit': 'GB/s',
        'frac': round(sim_gbs / PEAK_HBM_GBS, 4),
        'traffic': pmc_traffic('similarity', N_QUERIES, features=dim, nvox=nvox, kernel=sim_kernel),
        'kernel': sim_kernel, 'launches': int(sim_k_n), 'avg_launch_ms': round(sim_avg_ms, 5),
        'bytes_per_launch': sim_bytes,
        'cold': {'achieved': round(cold_gbs, 1), 'frac': round(cold_gbs / PEAK_HBM_GBS, 4), 'avg_launch_ms': round(cold_avg_ms, 5),
                 'launches': int(cold_n), 'note': 'the query rotating over three copies of the feature volume (603 MB): no launch finds its volume in the Infinity Cache'},
        'note': f'algorithmic bytes = Nvox * (2 D + C) = {nvox} * (2*{dim} + {n_classes}); achieved / frac: the query repeats over ONE resident '
                f'{nvox * 2 * dim / 1e6:.0f} MB volume, which the 256 MB Infinity Cache can serve in part -- "cold" is the HBM figure',
    }

    flop_slice = vit_flops(n_tokens, dim, depth, patch)
    slices_done = my_slices * args.steps
    if prof['mlp'][1] > 0:          # fused MLP: fc1 + fc2 in one kernel ...
        flop_slice['mlp'] = flop_slice.pop('gemm_fc1') + flop_slice.pop('gemm_fc2')
        if prof['gemm_proj'][1] == 0:   # ... block tail: with the attention projection in front of them
            flop_slice['mlp'] += flop_slice.pop('gemm_proj')
    flops = {k: v * slices_done for k, v in flop_slice.items()}
    kernels = {'attention': f"{vt._lib.kernel_name('attention')}<{args.dtype}>", 'gemm_qkv': f"{vt._lib.kernel_name('gemm_qkv') or 'gemm_kernel'}<{args.dtype}, qkv>",
               'gemm_fc1': f'gemm_kernel<{args.dtype}, fc1+gelu>', 'gemm_proj': f'gemm_rows_kernel<{args.dtype}, proj+ln>',
               'gemm_fc2': f'gemm_rows_kernel<{args.dtype}, fc2+ln>', 'gemm': f'gemm_kernel<{args.dtype}, kfeat>',
               'mlp': f"{vt._lib.kernel_name('mlp') or 'tail_fx_kernel'}<{args.dtype}>", 'patch_embed': 'patch_embed_kernel', 'layernorm': 'layernorm_kernel'}
    # the kernel with the largest share of the step
    dom = max((k for k in prof if k in flops), key=lambda k: prof[k][0])
    dom_ms, dom_launches = prof[dom]
    achieved = flops[dom] / (dom_ms * 1e-3) / 1e12 if dom_ms > 0 else 0.0
    # the dominant kernel's own operand type prices it: fp8 attention operands run against the 5 PF fp8 peak
    peak = PEAK_TFLOPS['fp8' if (dom == 'attention' and args.attention == 'fp8') else args.dtype]
    vit_ms = sum(v[0] for k, v in prof.items() if k != 'similarity')
    roofline = {
        'bound': 'mfma', 'achieved': round(achieved, 2), 'peak': peak, 'unit': 'TFLOP/s', 'frac': round(achieved / peak, 4),
        'traffic': pmc_traffic(dom, args.engine_batch if my_slices >= args.engine_batch else my_slices, tokens=n_tokens, heads=heads,
                               kernel=kernels[dom])
        if (dom == 'attention' and args.attention == '16bit') else None,
        'kernel': kernels[dom],
        'launches': int(dom_launches), 'avg_launch_ms': round(dom_ms / max(1, dom_launches), 4),
        'flop_per_launch': flops[dom] / max(1, dom_launches),
        'kernel_ms_rank0': {k: round(v[0], 2) for k, v in prof.items() if k != 'similarity'},
        'kernel_tflops_rank0': {k: round(flops[k] / (prof[k][0] * 1e-3) / 1e12, 1) for k in flops if prof[k][0] > 0},
        'kernel_ms_note': 'attention: events inside the timed steps; other classes: one extra untimed step x steps',
        'whole_vit_tflops': round(sum(flops.values()) / (vit_ms * 1e-3) / 1e12, 2) if vit_ms > 0 else 0.0,
    }

    # the second-largest kernel since round 3: everything behind the attention of a block in one launch (csrc/tail_fx.hip)
    roofline_tail = None
    if prof['mlp'][1] > 0 and prof['mlp'][0] > 0:
        t_ms, t_n = prof['mlp']
        t_rows = (args.engine_batch if my_slices >= args.engine_batch else my_slices) * n_tokens
        t_ach = flops['mlp'] / (t_ms * 1e-3) / 1e12
        tpeak = PEAK_TFLOPS[args.dtype]
        roofline_tail = {'bound': 'mfma', 'achieved': round(t_ach, 2), 'peak': tpeak, 'unit': 'TFLOP/s', 'frac': round(t_ach / tpeak, 4),
                         'traffic': pmc_traffic('block_tail', args.engine_batch if my_slices >= args.engine_batch else my_slices,
                                                tokens=n_tokens, features=dim),
                         'kernel': kernels['mlp'], 'launches': int(t_n), 'avg_launch_ms': round(t_ms / t_n, 4),
                         'flop_per_launch': flops['mlp'] / t_n,
                         'algorithmic_bytes_per_launch': int(t_rows * (dim * 2 + dim * 4 * 2 + dim * 2)),
                         'note': 'events around its launches in one extra untimed step; algorithmic bytes per token row: attention output in '
                                 '(2 D) + fp32 residual in and out (8 D) + 16-bit LayerNorm output (2 D); the 2.65 MB of weights every '
                                 'workgroup re-reads come from L2 / the Infinity Cache; traffic: profiles/pmc_block_tail.json'}

    # the third: the qkv projection of every block but the first, activation-stationary (csrc/gemm_as.hip).  2.65 GFLOP per MB it
    # must move: priced on both roofs.
    roofline_qkv = None
    if args.arch == 'vits8' and prof.get('gemm_qkv', (0, 0))[1] > 0 and prof['gemm_qkv'][0] > 0:
        q_ms, q_n = prof['gemm_qkv']
        q_rows = (args.engine_batch if my_slices >= args.engine_batch else my_slices) * n_tokens
        q_bytes = int(q_rows * (dim * 2 + 3 * dim * 2))
        q_ach = flops['gemm_qkv'] / (q_ms * 1e-3) / 1e12
        roofline_qkv = {'bound': 'mfma', 'achieved': round(q_ach, 2), 'peak': PEAK_TFLOPS[args.dtype], 'unit': 'TFLOP/s',
                        'frac': round(q_ach / PEAK_TFLOPS[args.dtype], 4),
                        'traffic': pmc_traffic('gemm_qkv', args.engine_batch if my_slices >= args.engine_batch else my_slices,
                                               tokens=n_tokens, features=dim),
                        'kernel': kernels['gemm_qkv'], 'launches': int(q_n), 'avg_launch_ms': round(q_ms / q_n, 4),
                        'flop_per_launch': flops['gemm_qkv'] / q_n, 'algorithmic_bytes_per_launch': q_bytes,
                        'hbm_frac_at_algorithmic_bytes': round(q_bytes / (q_ms / q_n * 1e-3) / 1e9 / PEAK_HBM_GBS, 4),
                        'note': 'events around its launches in one extra untimed step; algorithmic bytes per token row: LayerNorm output in '
                                '(2 D) + q, k, v out (6 D); traffic: profiles/pmc_gemm_qkv.json'}

    if rank == 0:
        out = {
            'metric': f'slices/sec ({"ViT-S/8" if args.arch == "vits8" else "ViT-B/8"}, {args.workload}^3 vol: feature volume + '
                      f'{N_QUERIES}-query similarity)',
            'value': round(args.steps * total_slices / elapsed, 2),
            'unit': 'slices/s', 'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
            'ms_per_step': round(elapsed / args.steps * 1e3, 2), 'higher_is_better': True,
            'scaling': 'strong', 'vs_baseline': None,      # every N runs the same 512^3 volume: total work fixed, sharded over the ranks
            'dtype': args.dtype if args.attention == '16bit' else f'{args.dtype} (attention operands fp8 e4m3)', 'data': 'synthetic',
            'rccl_ranks': torch.distributed.get_world_size() if world > 1 else 1,
            'config': {'workload': desc, 'volume': list(dvol.shape), 'slices_per_step': total_slices,
                       'image': [im_sz[0], im_sz[1]], 'tokens': n_tokens, 'feature_volume': [dim, *feat_out],
                       'engine_batch': args.engine_batch, 'weights': f'seeded synthetic {args.arch} (seed 0)',
                       'parallelism': f'slices sharded over {world} rank(s), one all-gather per axis' if world > 1 else 'single GPU',
                       'dist_backend': backend},
            'roofline': roofline,
            'roofline_similarity': roofline_sim,
            'roofline_block_tail': roofline_tail,
            'roofline_gemm_qkv': roofline_qkv,
            'similarity': {'ms': round(sim_ms, 3), 'queries': N_QUERIES,
                           'mvoxel_sim_per_s': round(nvox * N_QUERIES / 1e6 / (sim_ms * 1e-3), 1),
                           'mvoxel_per_s': round(nvox / 1e6 / (sim_ms * 1e-3), 1),
                           'queries_per_s': round(N_QUERIES / (sim_ms * 1e-3), 1),
                           'ms_maps_on_device': round(sim_dev_ms, 3),
                           'maps_bytes_to_host': int(n_classes * (dvol.shape[0] // 2) * (dvol.shape[1] // 2) * (dvol.shape[2] // 2)),
                           'note': 'ms: the reference API (uint8 maps returned as CPU tensors: ms - ms_maps_on_device is the D2H copy of '
                                   'maps_bytes_to_host over PCIe); ms_maps_on_device: keep_on_device=True'},
        }
        out['e2e_incl_upload'] = e2e
        if (world == 1 and args.arch == 'vits8' and args.attention == '16bit' and args.fos == FOS and args.workload == '512'
                and os.environ.get('VITTF_BENCH_EXTRAS', '1') == '1'):
            out.update(extra_configs(vt, torch, dvol, vol, feats, dev, args.dtype))
        if world == 1 and args.cpu_slices > 0:
            out['cpu_baseline'] = cpu_baseline(sd, vol, args.cpu_slices, im_sz, args.arch, feats.cpu(), ann, tuple(dvol.shape))
        else:
            out['cpu_baseline'] = None
        print(json.dumps(out), flush=True)
    if world > 1:
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()


if __name__ == '__main__':
    main()
