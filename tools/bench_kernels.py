#!/usr/bin/env python3
"""Per-kernel microbenchmark on the headline shapes (batch 32 slices of N = 4097 tokens, ViT-S/8), timed with
events on the launch stream; random data (zeros inflate MFMA clocks).  Usage: python tools/bench_kernels.py [what...]"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import vit_tf_amd as vt   # noqa: E402
from vit_tf_amd import _lib   # noqa: E402

TDT = {'bf16': torch.bfloat16, 'fp16': torch.float16}


def timeit(fn, reps=10, warm=3, settle_s=float(os.environ.get('SETTLE_S', '0.4'))):
    """Mean launch time by events on the launch stream, after `warm` calls AND at least settle_s seconds of back-to-back
    calls: the chip's clock settles under sustained load (a 13-launch measurement from idle read 10-15 % slow)."""
    import time
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < settle_s:
        for _ in range(10):
            fn()
        torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps


def main():
    what = set(sys.argv[1:]) or {'attn', 'gemm', 'qkv', 'tail', 'ln', 'sim'}
    lib = _lib.load()
    dev = torch.device('cuda', 0)
    dt = os.environ.get('DT', 'fp16')
    d = int(os.environ.get('D', '384'))                    # D=768: the ViT-B/8 shapes (BASELINE configs[3])
    batch, tokens, heads = int(os.environ.get('BATCH', '32')), int(os.environ.get('TOKENS', '4097')), d // 64
    rows = batch * tokens
    g = torch.Generator(device='cpu').manual_seed(0)
    if 'attn' in what:
        qkv = torch.randn(rows, 3 * d, generator=g)
        qkv[:, :2 * d] *= 1.5
        pre = int(os.environ.get('PRE', '1'))
        if pre:
            qkv[:, :d] *= 0.125 * 1.4426950408889634
        qkv = qkv.to(TDT[dt]).to(dev)
        out = torch.empty(rows, d, dtype=TDT[dt], device=dev)
        ms = timeit(lambda: _lib.check(lib.vittf_attention(_lib.ptr(qkv), _lib.ptr(out), batch, tokens, heads, _lib.DTYPES[dt], pre, _lib.stream_ptr())))
        fl = batch * 4 * tokens * tokens * d
        print(f'attention  (q_prescaled={pre}) batch {batch} N {tokens}: {ms:.3f} ms  {fl / ms / 1e9:.1f} TFLOP/s  ({fl / ms / 1e9 / 25:.1f} % of 2.5 PF)')
    if 'attn8' in what:      # the fp8 attention path (absmax + quantise + attention), ViT-S and ViT-B head counts
        for hh in (6, 12):
            qkv = torch.randn(rows, 3 * hh * 64, generator=g)
            qkv[:, :2 * hh * 64] *= 1.5
            qkv[:, :hh * 64] *= 0.125 * 1.4426950408889634
            qkv = qkv.to(TDT[dt]).to(dev)
            out = torch.empty(rows, hh * 64, dtype=TDT[dt], device=dev)
            ws = torch.empty(lib.vittf_attention_fp8_workspace_bytes(batch, tokens, hh), dtype=torch.uint8, device=dev)
            ms = timeit(lambda: _lib.check(lib.vittf_attention_fp8(_lib.ptr(qkv), _lib.ptr(out), batch, tokens, hh, _lib.DTYPES[dt], _lib.ptr(ws), ws.numel(), _lib.stream_ptr())))
            fl = batch * 4 * tokens * tokens * hh * 64
            ms16 = timeit(lambda: _lib.check(lib.vittf_attention(_lib.ptr(qkv), _lib.ptr(out), batch, tokens, hh, _lib.DTYPES[dt], 1, _lib.stream_ptr())))
            print(f'attention fp8 ({hh} heads) batch {batch} N {tokens}: {ms:.3f} ms  {fl / ms / 1e9:.1f} TFLOP/s;  16-bit kernel on the same tensors: {ms16:.3f} ms')
    if 'gemm' in what:
        for name, n, k, epi in (('qkv', 3 * d, d, 0), ('qkv+qscale', 3 * d, d, 4), ('proj+res', d, d, 2), ('fc1+gelu', 4 * d, d, 1), ('fc2+res', d, 4 * d, 2), ('kfeat', d, d, 3)):
            a = torch.randn(rows, k, generator=g).to(TDT[dt]).to(dev)
            w = (torch.randn(n, k, generator=g) / k ** 0.5).to(TDT[dt]).to(dev)
            bias = torch.randn(n, generator=g).to(dev)
            o = torch.zeros(rows, n, dtype=torch.float32 if epi == 2 else TDT[dt], device=dev)
            ms = timeit(lambda: _lib.check(lib.vittf_gemm(_lib.ptr(a), _lib.ptr(w), _lib.ptr(bias), _lib.ptr(o), rows, n, k, epi, tokens, _lib.DTYPES[dt], _lib.stream_ptr())))
            fl = 2 * rows * n * k
            byts = rows * k * 2 + n * k * 2 + rows * n * (8 if epi == 2 else 2)
            print(f'gemm {name:9s} [{rows}x{k}]x[{n}x{k}]^T: {ms:.3f} ms  {fl / ms / 1e9:.1f} TFLOP/s   {byts / ms / 1e6:.0f} GB/s algorithmic')
            if epi == 2:   # the same GEMM with the following LayerNorm in its epilogue (16-bit h written as well)
                lg = torch.ones(n, device=dev); lb = torch.zeros(n, device=dev)
                h = torch.empty(rows, n, dtype=TDT[dt], device=dev)
                ms = timeit(lambda: _lib.check(lib.vittf_gemm_residual_ln(_lib.ptr(a), _lib.ptr(w), _lib.ptr(bias), _lib.ptr(o), rows, n, k, _lib.DTYPES[dt], _lib.ptr(lg), _lib.ptr(lb), 1e-6, _lib.ptr(h), _lib.stream_ptr())))
                byts += rows * n * 2
                print(f'gemm {name + "+ln":9s} [{rows}x{k}]x[{n}x{k}]^T: {ms:.3f} ms  {fl / ms / 1e9:.1f} TFLOP/s   {byts / ms / 1e6:.0f} GB/s algorithmic')
    if 'stock' in what:   # the same shapes on the stock ROCm libraries (hipBLASLt, SDPA): what unmodified torch would do here
        import torch.nn.functional as F
        for name, n, k in (('qkv', 3 * d, d), ('proj', d, d), ('fc1', 4 * d, d), ('fc2', d, 4 * d)):
            a = torch.randn(rows, k, generator=g).to(TDT[dt]).to(dev)
            w = (torch.randn(n, k, generator=g) / k ** 0.5).to(TDT[dt]).to(dev)
            bias = torch.randn(n, generator=g).to(TDT[dt]).to(dev)
            ms = timeit(lambda: F.linear(a, w, bias))
            print(f'stock linear {name:5s} [{rows}x{k}]x[{n}x{k}]^T (16-bit out, no epilogue): {ms:.3f} ms  {2 * rows * n * k / ms / 1e9:.1f} TFLOP/s')
        q = torch.randn(batch, heads, tokens, 64, generator=g).to(TDT[dt]).to(dev)
        kk = torch.randn(batch, heads, tokens, 64, generator=g).to(TDT[dt]).to(dev)
        v = torch.randn(batch, heads, tokens, 64, generator=g).to(TDT[dt]).to(dev)
        try:
            ms = timeit(lambda: F.scaled_dot_product_attention(q, kk, v))
            fl = batch * 4 * tokens * tokens * d
            print(f'stock SDPA batch {batch} N {tokens}: {ms:.3f} ms  {fl / ms / 1e9:.1f} TFLOP/s')
        except Exception as e:  # noqa: BLE001
            print('stock SDPA failed:', e)
    if 'bilateral' in what:   # predict_ntf.py:73-96 for one class at the BASELINE configs[4] size (512^3 volume -> 256^3 maps)
        import time
        size = int(os.environ.get('BLS_SIZE', '512'))
        vol, lab = vt.ct_like_volume(size, seed=0)
        vol = vol.float().to(dev)
        zz, yy, xx = torch.meshgrid(*[torch.linspace(-1, 1, 64)] * 3, indexing='ij')
        sim = (torch.exp(-6 * ((zz - 0.1) ** 2 + yy ** 2 + (xx + 0.2) ** 2)) + 0.05 * torch.rand(64, 64, 64, generator=g)).to(dev)
        shape = (size // 2,) * 3
        info = {}
        vt.bilateral.refine_similarity(sim, vol, shape, info=info)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(3):
            out = vt.bilateral.refine_similarity(sim, vol, shape)
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / 3 * 1e3
        print(f'bilateral refine {size}^3 volume -> {shape[0]}^3 map: {ms:.1f} ms  ({info["voxels"]} voxels in the crop box, {info["vertices"]} vertices)')
        if os.environ.get('BLS_CPU', '0') == '1':
            from oracle import bilateral as obil
            t0 = time.perf_counter()
            ref = obil.refine_similarity(sim.cpu(), vol.cpu(), shape)
            cpu_s = time.perf_counter() - t0
            err = (out.cpu() - ref).abs()
            print(f'  CPU restatement (numpy fp64, 1 process): {cpu_s:.1f} s; GPU vs CPU: median |diff| {float(err.median()):.2e}, '
                  f'voxels off by > 1e-3: {float((err > 1e-3).float().mean()):.2e}')
    if 'labels' in what:   # sampler candidate masks + confusion matrix at the BASELINE configs[4] size (512^3 uint8 labels)
        size = int(os.environ.get('LAB_SIZE', '512'))
        _, lab = vt.ct_like_volume(size, seed=0)
        dl = vt.samplers.device_labels(lab)
        ms = timeit(lambda: vt.samplers.surface_shell(dl, 4, class_id=1), reps=5, warm=2)
        print(f'surface shell {size}^3 (cube erosion + 6-neighbour erosion + xor): {ms:.3f} ms  {4 * dl.numel() / ms / 1e6:.0f} GB/s algorithmic (2 reads + 2 writes per voxel)')
        pred = torch.where(torch.rand(dl.shape, device=dev) < 0.8, dl, torch.zeros_like(dl))
        k = int(dl.max().item()) + 1
        counts = torch.empty(k * k + 1, dtype=torch.int64, device=dev)
        ms = timeit(lambda: _lib.check(lib.vittf_confusion_matrix(_lib.ptr(dl), _lib.ptr(pred), dl.numel(), k, _lib.ptr(counts), _lib.stream_ptr())), reps=5, warm=2)
        print(f'confusion matrix {size}^3, {k} classes: {ms:.3f} ms  {2 * dl.numel() / ms / 1e6:.0f} GB/s')
        if os.environ.get('LAB_CPU', '0') == '1':
            import time
            from oracle import samplers as osmp
            t0 = time.perf_counter(); ref = osmp.surface_shell(lab.numpy() == 1); cpu_s = time.perf_counter() - t0
            same = bool((vt.samplers.surface_shell(dl, 4, class_id=1).cpu().numpy().astype(bool) == ref).all())
            print(f'  scipy binary_erosion x2 + xor on the host: {cpu_s:.1f} s; identical: {same}')
    if 'qkv' in what:      # the activation-stationary qkv projection (csrc/gemm_as.hip), K = 384, N = 1152
        hh = torch.randn(rows, d, generator=g).to(TDT[dt]).to(dev)
        wq = (torch.randn(3 * d, d, generator=g) / d ** 0.5).to(TDT[dt]).to(dev)
        wpk = vt.weights.pack_row_images(wq[None])[0].contiguous()
        bq = torch.randn(3 * d, generator=g).to(dev)
        out = torch.empty(rows, 3 * d, dtype=TDT[dt], device=dev)
        ctr = torch.zeros(1, dtype=torch.int32, device=dev)
        ms = timeit(lambda: _lib.check(lib.vittf_gemm_as(_lib.ptr(hh), _lib.ptr(wpk), _lib.ptr(bq), _lib.ptr(out), rows, 3 * d, d,
                                                          _lib.EPI_BIAS_QKV, _lib.DTYPES[dt], _lib.ptr(ctr), _lib.stream_ptr())))
        fl = 2 * rows * d * 3 * d
        print(f'qkv projection [{rows}x{d}] x [{3 * d}x{d}]^T: {ms:.3f} ms  {fl / ms / 1e9:.1f} TFLOP/s  '
              f'({(rows * d * 2 + rows * 3 * d * 2) / ms / 1e6:.0f} GB/s algorithmic)')
    if 'tail' in what:
        aa = torch.randn(rows, d, generator=g).to(TDT[dt]).to(dev)
        wp = (torch.randn(d, d, generator=g) / d ** 0.5).to(TDT[dt]).to(dev)
        w1 = (torch.randn(4 * d, d, generator=g) / d ** 0.5).to(TDT[dt]).to(dev)
        w2 = (torch.randn(d, 4 * d, generator=g) / (4 * d) ** 0.5).to(TDT[dt]).to(dev)
        tpk = vt.weights.pack_block_tail_weights(wp[None], w1[None], w2[None])[0].contiguous()
        bp = torch.randn(d, generator=g).to(dev); b1 = torch.randn(4 * d, generator=g).to(dev); b2 = torch.randn(d, generator=g).to(dev)
        x = torch.zeros(rows, d, device=dev)
        lg = torch.ones(d, device=dev); lb = torch.zeros(d, device=dev)
        hn = torch.empty(rows, d, dtype=TDT[dt], device=dev)
        ctr = torch.zeros(1, dtype=torch.int32, device=dev)
        ms = timeit(lambda: _lib.check(lib.vittf_block_tail(_lib.ptr(aa), _lib.ptr(tpk), _lib.ptr(bp), _lib.ptr(lg), _lib.ptr(lb), _lib.ptr(b1),
                                                             _lib.ptr(b2), _lib.ptr(x), rows, d, _lib.DTYPES[dt], _lib.ptr(lg), _lib.ptr(lb), 1e-6,
                                                             _lib.ptr(hn), _lib.ptr(ctr), _lib.stream_ptr())))
        fl = 18 * rows * d * d
        print(f'block tail (proj + ln + mlp + ln) [{rows}x{d}]: {ms:.3f} ms  {fl / ms / 1e9:.1f} TFLOP/s  ({ms * 32 / batch:.4f} ms per 32 slices)')
    if 'ln' in what:
        x = torch.randn(rows, d, generator=g).to(dev)
        w = torch.ones(d, device=dev); b = torch.zeros(d, device=dev)
        y = torch.empty(rows, d, dtype=TDT[dt], device=dev)
        ms = timeit(lambda: _lib.check(lib.vittf_layernorm(_lib.ptr(x), _lib.ptr(w), _lib.ptr(b), _lib.ptr(y), rows, d, 1e-6, _lib.DTYPES[dt], _lib.stream_ptr())))
        print(f'layernorm [{rows}x{d}]: {ms:.3f} ms  {rows * d * 6 / ms / 1e6:.0f} GB/s')
    if 'sim' in what:
        import numpy as np
        feat = torch.randn(384, 64, 64, 64, generator=g).half().to(dev)
        for na in (1, 16, 64):
            ann = {'a': torch.randint(0, 256, (na, 3), generator=g)}
            vol = np.zeros((256, 256, 256), np.float32)
            ms = timeit(lambda: vt.compute_similarities(vol, feat, ann), reps=5, warm=2)
            print(f'similarity 64^3x384, A={na}: {ms:.3f} ms end-to-end  ({feat.numel() * 2 / ms / 1e6:.0f} GB/s of feature bytes, {262144 * na / ms / 1e3:.0f} Mvoxel-sim/s)')
        fd = feat.to(dev)
        ms = timeit(lambda: vt.similarity.voxel_norms(fd), reps=5, warm=2)
        print(f'voxel norms 64^3x384: {ms:.3f} ms  ({feat.numel() * 2 / ms / 1e6:.0f} GB/s)')
        # BASELINE configs[4]: 1024 annotations for each of 5 classes (VITTF_SIM_MFMA=0 keeps the VALU kernel)
        ann = {f'c{i}': torch.randint(0, 256, (1024, 3), generator=g) for i in range(5)}
        ms = timeit(lambda: vt.compute_similarities(vol, feat, ann), reps=3, warm=1)
        print(f'similarity 64^3x384, A=5x1024 (VITTF_SIM_MFMA={os.environ.get("VITTF_SIM_MFMA", "1")}): {ms:.2f} ms end-to-end  '
              f'({262144 * 5120 / ms / 1e3:.0f} Mvoxel-sim/s, {2 * 262144 * 5120 * 384 / ms / 1e9:.1f} TFLOP/s of dot products)')


if __name__ == '__main__':
    main()
