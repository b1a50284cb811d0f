"""GPU: each libvittf kernel, called through the C ABI, against an fp64 / oracle restatement of the op it
replaces.  Tolerances are stated per test; integer / byte / fp16-pooling work is bit-exact."""
import ctypes as C

import numpy as np
import pytest
import torch
import torch.nn.functional as F

import vit_tf_amd as vt
from vit_tf_amd import _lib
from oracle import dino_vit, feature_volume as ofv, similarity as osim
from helpers import load_golden, rel_fro, max_abs, TINY_ARCH

pytestmark = pytest.mark.gpu

TDT = {'bf16': torch.bfloat16, 'fp16': torch.float16}
EPS = {'bf16': 2.0 ** -8, 'fp16': 2.0 ** -11}      # half-ulp relative rounding error of the 16-bit type


def gen(seed):
    return torch.Generator().manual_seed(seed)


# ------------------------------------------------------------------------------------------ LayerNorm
@pytest.mark.parametrize('dt', ['bf16', 'fp16'])
@pytest.mark.parametrize('rows,d', [(1, 128), (5, 384), (1001, 384), (130, 768)])
def test_layernorm(gpu, dt, rows, d):
    lib = _lib.load()
    g = gen(rows + d)
    x = torch.randn(rows, d, generator=g) * 3.0 + torch.randn(rows, 1, generator=g) * 5.0
    w = 1.0 + 0.1 * torch.randn(d, generator=g)
    b = 0.1 * torch.randn(d, generator=g)
    ref = F.layer_norm(x.double(), (d,), w.double(), b.double(), eps=1e-6)
    xd, wd, bd = x.to(gpu), w.to(gpu), b.to(gpu)
    y = torch.zeros(rows, d, dtype=TDT[dt], device=gpu)
    _lib.check(lib.vittf_layernorm(_lib.ptr(xd), _lib.ptr(wd), _lib.ptr(bd), _lib.ptr(y), rows, d, 1e-6, _lib.DTYPES[dt],
                                   _lib.stream_ptr()))
    got = y.float().cpu().double()
    # fp32 statistics + one rounding to the 16-bit type
    assert ((got - ref).abs() <= EPS[dt] * ref.abs() * 1.01 + 2e-5).all()


# ------------------------------------------------------------------------------------------ GEMM
def _gemm_inputs(rows, n, k, dt, seed):
    g = gen(seed)
    a = (torch.randn(rows, k, generator=g)).to(TDT[dt])
    w = (torch.randn(n, k, generator=g) / k ** 0.5).to(TDT[dt])
    bias = 0.2 * torch.randn(n, generator=g)
    ref = a.double() @ w.double().t() + bias.double()
    return a, w, bias, ref


@pytest.mark.parametrize('dt', ['bf16', 'fp16'])
@pytest.mark.parametrize('rows,n,k', [(1, 128, 64), (100, 128, 128), (128, 384, 384), (300, 1152, 384),
                                      (257, 384, 1536), (8194, 1536, 384), (19205, 1152, 384),
                                      (1, 256, 768), (300, 2304, 768), (4097, 3072, 768), (513, 768, 3072), (8194, 256, 832), (700, 128, 768)])   # K >= 768, N % 256 == 0: gemm_pp.hip
def test_gemm_bias_and_gelu(gpu, dt, rows, n, k):
    lib = _lib.load()
    a, w, bias, ref = _gemm_inputs(rows, n, k, dt, rows + n + k)
    ad, wd, bd = a.to(gpu), w.to(gpu), bias.to(gpu)
    for epi, refv in ((_lib.EPI_BIAS, ref), (_lib.EPI_BIAS_GELU, F.gelu(ref))):
        out = torch.full((rows + 3, n), 7.0, dtype=TDT[dt], device=gpu)       # 3 guard rows
        _lib.check(lib.vittf_gemm(_lib.ptr(ad), _lib.ptr(wd), _lib.ptr(bd), _lib.ptr(out), rows, n, k, epi, 0,
                                  _lib.DTYPES[dt], _lib.stream_ptr()))
        got = out.float().cpu().double()
        assert (got[rows:] == 7.0).all(), 'wrote past the last row'
        # fp32 accumulation (error ~ 1e-6 * sqrt(k)) + one rounding of the output
        assert ((got[:rows] - refv).abs() <= EPS[dt] * refv.abs() * 1.01 + 3e-5 * k ** 0.5).all()


@pytest.mark.parametrize('case', ['bias', 'gelu', 'residual', 'kfeat'])
def test_gemm_pp_many_tiles_per_workgroup_same_bits_as_one_tile_launches(gpu, case):
    """csrc/gemm_pp.hip is persistent: with more 256 x 256 tiles than CUs a workgroup runs several tiles back to back -- the ring
    never drains, the next tile's first stages are requested under the counted waits of this one, the epilogue's stores stay in
    flight under the next K loop.  Every other gemm_pp test shape has fewer tiles than an MI355X has CUs, so this path is checked
    here per element: one launch with 303 .. 2709 tiles against the SAME rows computed by launches of at most 255 tiles (one tile
    per workgroup), bit for bit, plus fp64."""
    lib = _lib.load()
    dt = 'fp16'
    tokens = 4097
    rows, n, k, epi, chunk = {'bias': (256 * 300 + 33, 2304, 768, _lib.EPI_BIAS, 256 * 28),
                              'gelu': (256 * 40, 3072, 768, _lib.EPI_BIAS_GELU, 256 * 21),
                              'residual': (256 * 100 + 7, 768, 3072, _lib.EPI_BIAS_RESIDUAL, 256 * 85),
                              'kfeat': (20 * tokens, 768, 768, _lib.EPI_KFEAT, 5 * tokens)}[case]
    assert (rows + 255) // 256 * (n // 256) > 256 + 32 and (chunk + 255) // 256 * (n // 256) <= 255
    g = gen(rows + n)
    a = torch.randn(rows, k, generator=g).to(TDT[dt]).to(gpu)
    w = (torch.randn(n, k, generator=g) / k ** 0.5).to(TDT[dt]).to(gpu)
    bias = (0.2 * torch.randn(n, generator=g)).to(gpu)
    out_rows = rows - rows // tokens if case == 'kfeat' else rows
    odt = torch.float32 if case == 'residual' else (torch.float16 if case == 'kfeat' else TDT[dt])
    x0 = (torch.randn(out_rows + 3, n, generator=g) * 4).to(odt).to(gpu) if case == 'residual' else torch.full((out_rows + 3, n), 7.0, dtype=odt, device=gpu)
    big = x0.clone()
    _lib.check(lib.vittf_gemm(_lib.ptr(a), _lib.ptr(w), _lib.ptr(bias), _lib.ptr(big), rows, n, k, epi, tokens, _lib.DTYPES[dt], _lib.stream_ptr()))
    small = x0.clone()
    for r0 in range(0, rows, chunk):
        r1 = min(rows, r0 + chunk)
        o0 = r0 - r0 // tokens if case == 'kfeat' else r0
        _lib.check(lib.vittf_gemm(_lib.ptr(a[r0:]), _lib.ptr(w), _lib.ptr(bias), _lib.ptr(small[o0:]), r1 - r0, n, k, epi, tokens, _lib.DTYPES[dt],
                                  _lib.stream_ptr()))
    torch.cuda.synchronize()
    assert torch.equal(big[out_rows:], x0[out_rows:]), 'wrote past the last row'
    assert torch.equal(big, small), f'{int((big != small).sum())} values differ between one launch and one-tile-per-workgroup launches'
    ref = a.double() @ w.double().t() + bias.double()
    if case == 'gelu':
        ref = F.gelu(ref)
    if case == 'kfeat':
        ref = ref.view(20, tokens, n)[:, 1:].reshape(-1, n)
    if case == 'residual':
        ref = ref + x0[:rows].double()
    assert rel_fro(big[:out_rows].double(), ref) <= EPS[dt]


@pytest.mark.parametrize('dt', ['bf16', 'fp16'])
@pytest.mark.parametrize('rows,n,k', [(3, 128, 64), (515, 384, 384), (200, 384, 1536), (33000, 384, 384), (1030, 384, 1536), (256, 384, 64),
                                       (515, 768, 768), (200, 768, 3072), (33000, 768, 768), (129, 768, 64)])
def test_gemm_residual(gpu, dt, rows, n, k):
    lib = _lib.load()
    a, w, bias, ref = _gemm_inputs(rows, n, k, dt, rows * 3 + n)
    x0 = torch.randn(rows + 2, n, generator=gen(5)) * 4.0
    xd, ad, wd, bd = x0.to(gpu), a.to(gpu), w.to(gpu), bias.to(gpu)      # keep the device buffers alive
    _lib.check(lib.vittf_gemm(_lib.ptr(ad), _lib.ptr(wd), _lib.ptr(bd), _lib.ptr(xd), rows, n, k,
                              _lib.EPI_BIAS_RESIDUAL, 0, _lib.DTYPES[dt], _lib.stream_ptr()))
    got = xd.cpu().double()
    assert torch.equal(got[rows:], x0[rows:].double())
    assert ((got[:rows] - (x0[:rows].double() + ref)).abs() <= 1e-5 * k ** 0.5 + 1e-6 * ref.abs()).all()


@pytest.mark.parametrize('dt', ['bf16', 'fp16'])
@pytest.mark.parametrize('rows,n,epi', [(1, 1152, 'qkv'), (255, 1152, 'qkv'), (4097, 1152, 'qkv'), (256 * 300 + 77, 1152, 'qkv'),
                                        (4097, 384, 'bias'), (1000, 1536, 'bias')])
def test_gemm_as_same_bits_as_tiled_kernel(gpu, dt, rows, n, epi):
    """The activation-stationary K = 384 GEMM (csrc/gemm_as.hip: a wave keeps its rows as 24 B fragments, the weights stream
    through the ring) forms every output as vittf_gemm's kernels do -- one ascending-k chain from zero, then
    (acc + bias) * scale -- so the outputs are bit-equal; more row tiles than workgroups (the stream wraps, tiles come from the
    counter), a partial last tile, guard rows untouched; against fp64 as well."""
    lib = _lib.load()
    k = 384
    g = gen(rows + n)
    a = torch.randn(rows, k, generator=g).to(TDT[dt]).to(gpu)
    w = (1.3 * torch.randn(n, k, generator=g) / k ** 0.5).to(TDT[dt]).to(gpu)
    bias = (0.3 * torch.randn(n, generator=g)).to(gpu)
    e = _lib.EPI_BIAS_QKV if epi == 'qkv' else _lib.EPI_BIAS
    ref = torch.full((rows + GUARD_ROWS, n), 5.0, dtype=TDT[dt], device=gpu)
    _lib.check(lib.vittf_gemm(_lib.ptr(a), _lib.ptr(w), _lib.ptr(bias), _lib.ptr(ref), rows, n, k, e, 0, _lib.DTYPES[dt], _lib.stream_ptr()))
    wpk = vt.weights.pack_row_images(w[None])[0]
    assert wpk.shape == (n // 32, 12288)
    out = torch.full((rows + GUARD_ROWS, n), 5.0, dtype=TDT[dt], device=gpu)
    ctr = torch.full((1,), 777, dtype=torch.int32, device=gpu)
    assert lib.vittf_gemm_as_workspace_bytes() <= 4
    _lib.check(lib.vittf_gemm_as(_lib.ptr(a), _lib.ptr(wpk), _lib.ptr(bias), _lib.ptr(out), rows, n, k, e, _lib.DTYPES[dt], _lib.ptr(ctr),
                                 _lib.stream_ptr()))
    torch.cuda.synchronize()
    assert (out[rows:].float() == 5.0).all(), 'wrote past the last row'
    assert torch.equal(out[:rows], ref[:rows]), f'{int((out[:rows] != ref[:rows]).sum())} values differ from vittf_gemm'
    want = a.double() @ w.double().t() + bias.double()
    if epi == 'qkv':
        want[:, :n // 3] *= QSCALE
    assert rel_fro(out[:rows].double(), want) <= EPS[dt]
    # refused: other k, odd column counts, a missing counter
    assert lib.vittf_gemm_as(_lib.ptr(a), _lib.ptr(wpk), _lib.ptr(bias), _lib.ptr(out), rows, n, 768, e, _lib.DTYPES[dt], _lib.ptr(ctr), _lib.stream_ptr()) == -1
    assert lib.vittf_gemm_as(_lib.ptr(a), _lib.ptr(wpk), _lib.ptr(bias), _lib.ptr(out), rows, n + 32, k, _lib.EPI_BIAS, _lib.DTYPES[dt], _lib.ptr(ctr), _lib.stream_ptr()) == -1
    assert lib.vittf_gemm_as(_lib.ptr(a), _lib.ptr(wpk), _lib.ptr(bias), _lib.ptr(out), rows, n, k, e, _lib.DTYPES[dt], None, _lib.stream_ptr()) == -1


@pytest.mark.parametrize('dt', ['bf16', 'fp16'])
def test_gemm_kfeat_drops_cls_rows(gpu, dt):
    lib = _lib.load()
    tokens, batch, n, k = 17, 9, 128, 128
    rows = tokens * batch
    a, w, bias, ref = _gemm_inputs(rows, n, k, dt, 99)
    out = torch.full((batch * (tokens - 1) + 1, n), 7.0, dtype=torch.float16, device=gpu)
    ad, wd, bd = a.to(gpu), w.to(gpu), bias.to(gpu)
    _lib.check(lib.vittf_gemm(_lib.ptr(ad), _lib.ptr(wd), _lib.ptr(bd), _lib.ptr(out), rows, n, k,
                              _lib.EPI_KFEAT, tokens, _lib.DTYPES[dt], _lib.stream_ptr()))
    got = out.float().cpu().double()
    want = ref.view(batch, tokens, n)[:, 1:].reshape(-1, n)
    assert (got[-1] == 7.0).all()
    assert ((got[:-1] - want).abs() <= EPS['fp16'] * want.abs() * 1.01 + 3e-5 * k ** 0.5).all()


def test_gemm_rejects_bad_shapes(gpu):
    lib = _lib.load()
    t = torch.zeros(256, 256, dtype=torch.float16, device=gpu)
    f = torch.zeros(256, dtype=torch.float32, device=gpu)
    for n, k in ((100, 64), (128, 60)):
        assert lib.vittf_gemm(_lib.ptr(t), _lib.ptr(t), _lib.ptr(f), _lib.ptr(t), 4, n, k, 0, 0, 1, _lib.stream_ptr()) == -1
    assert lib.vittf_gemm(None, _lib.ptr(t), _lib.ptr(f), _lib.ptr(t), 4, 128, 64, 0, 0, 1, _lib.stream_ptr()) == -1


# Rows behind the last valid one in x / h_out of the fused kernels: a partial tile's stores carry row offsets of 8, 16 and 24
# rows in the buffer instructions' soffset operand; on gfx950 voffset + soffset IS range-checked against the descriptor (the
# rows=128*300+77 case puts such a store on a guard row and leaves it untouched) -- 32 guard rows let every shape detect it.
GUARD_ROWS = 32


@pytest.mark.parametrize('dt', ['bf16', 'fp16'])
@pytest.mark.parametrize('rows', [1, 130, 4097, 128 * 300 + 77])
def test_block_tail(gpu, dt, rows):
    """proj + residual + norm2 + MLP + residual + the next norm1 in one launch (vittf_block_tail, csrc/tail_fx.hip) against fp64
    (norm2's output and the hidden activation rounded once to the 16-bit type, as every path does) and against the three GEMM
    launches it replaces.  (Round 5: the role-split kernel was bit-equal to round 3's one-wave kernel on all of these shapes
    before that kernel was removed: commit 89da172, profiles/r05b.)"""
    lib = _lib.load()
    d = 384
    g = gen(rows + 5)
    a = torch.randn(rows, d, generator=g).to(TDT[dt])
    wp = (torch.randn(d, d, generator=g) / d ** 0.5).to(TDT[dt])
    bp = 0.3 * torch.randn(d, generator=g)
    w1 = (torch.randn(4 * d, d, generator=g) / d ** 0.5).to(TDT[dt])
    b1 = 0.3 * torch.randn(4 * d, generator=g)
    w2 = (torch.randn(d, 4 * d, generator=g) / (4 * d) ** 0.5).to(TDT[dt])
    b2 = 0.3 * torch.randn(d, generator=g)
    g2, e2 = 1.0 + 0.2 * torch.randn(d, generator=g), 0.1 * torch.randn(d, generator=g)
    g1, e1 = 1.0 + 0.2 * torch.randn(d, generator=g), 0.1 * torch.randn(d, generator=g)
    x0 = torch.randn(rows + GUARD_ROWS, d, generator=g) * 3
    ctr = torch.full((1,), 12345, dtype=torch.int32, device=gpu)      # caller-owned tile counter (zeroed by the call itself)
    ad, wpd, bpd, w1d, b1d, w2d, b2d, g2d, e2d, g1d, e1d = (t.to(gpu) for t in (a, wp, bp, w1, b1, w2, b2, g2, e2, g1, e1))
    # fp64 reference
    x1 = x0[:rows].to(gpu).double() + ad.double() @ wpd.double().t() + bpd.double()
    hn = F.layer_norm(x1, (d,), g2d.double(), e2d.double(), 1e-6).to(TDT[dt]).double()
    hid = F.gelu(hn @ w1d.double().t() + b1d.double()).to(TDT[dt]).double()
    ref = (x1 + hid @ w2d.double().t() + b2d.double()).cpu()
    wpk = vt.weights.pack_block_tail_weights(wpd[None], w1d[None], w2d[None])[0].contiguous()
    assert wpk.shape == (112, 12288)
    xd = x0.to(gpu)
    hout = torch.full((rows + GUARD_ROWS, d), 7.0, dtype=TDT[dt], device=gpu)
    _lib.check(lib.vittf_block_tail(_lib.ptr(ad), _lib.ptr(wpk), _lib.ptr(bpd), _lib.ptr(g2d), _lib.ptr(e2d), _lib.ptr(b1d), _lib.ptr(b2d),
                                    _lib.ptr(xd), rows, d, _lib.DTYPES[dt], _lib.ptr(g1d), _lib.ptr(e1d), 1e-6, _lib.ptr(hout),
                                    _lib.ptr(ctr), _lib.stream_ptr()))
    got = xd.cpu().double()
    assert torch.equal(got[rows:], x0[rows:].double()), 'wrote past the last row'
    assert (hout[rows:].float() == 7.0).all(), 'wrote past the last row of h'
    # a norm2 output / hidden unit on a rounding boundary may round the other way than in fp64: compare in norm
    assert rel_fro(got[:rows] - x0[:rows].double(), ref - x0[:rows].double()) <= EPS[dt] / 2
    assert ((got[:rows] - ref).abs() <= 6 * EPS[dt] + 1e-4).all()
    want = F.layer_norm(got[:rows], (d,), g1.double(), e1.double(), 1e-6)
    assert ((hout[:rows].cpu().double() - want).abs() <= 2 * EPS[dt] * (1 + want.abs())).all()
    # the launches it replaces: proj + residual + norm2, fc1 + GELU, fc2 + residual + the next norm1
    x2 = x0.to(gpu)
    h2 = torch.zeros(rows, d, dtype=TDT[dt], device=gpu)
    _lib.check(lib.vittf_gemm_residual_ln(_lib.ptr(ad), _lib.ptr(wpd), _lib.ptr(bpd), _lib.ptr(x2), rows, d, d, _lib.DTYPES[dt],
                                          _lib.ptr(g2d), _lib.ptr(e2d), 1e-6, _lib.ptr(h2), _lib.stream_ptr()))
    gbuf = torch.zeros(rows, 4 * d, dtype=TDT[dt], device=gpu)
    _lib.check(lib.vittf_gemm(_lib.ptr(h2), _lib.ptr(w1d), _lib.ptr(b1d), _lib.ptr(gbuf), rows, 4 * d, d, _lib.EPI_BIAS_GELU, 0,
                              _lib.DTYPES[dt], _lib.stream_ptr()))
    h3 = torch.zeros(rows, d, dtype=TDT[dt], device=gpu)
    _lib.check(lib.vittf_gemm_residual_ln(_lib.ptr(gbuf), _lib.ptr(w2d), _lib.ptr(b2d), _lib.ptr(x2), rows, d, 4 * d, _lib.DTYPES[dt],
                                          _lib.ptr(g1d), _lib.ptr(e1d), 1e-6, _lib.ptr(h3), _lib.stream_ptr()))
    assert rel_fro(xd[:rows].double() - x0[:rows].to(gpu).double(), x2[:rows].double() - x0[:rows].to(gpu).double()) <= EPS[dt] / 2
    assert rel_fro(hout[:rows].float(), h3.float()) <= 2 * EPS[dt]
    # misaligned rows are refused, not mis-read (every access is 16 bytes wide)
    assert lib.vittf_block_tail(C.c_void_p(ad.data_ptr() + 2), _lib.ptr(wpk), _lib.ptr(bpd), _lib.ptr(g2d), _lib.ptr(e2d), _lib.ptr(b1d), _lib.ptr(b2d),
                                _lib.ptr(xd), rows, d, _lib.DTYPES[dt], None, None, 1e-6, None, _lib.ptr(ctr), _lib.stream_ptr()) == -1
    # without the LayerNorm on the way out: the same residual rows, nothing else written
    x3 = x0.to(gpu)
    _lib.check(lib.vittf_block_tail(_lib.ptr(ad), _lib.ptr(wpk), _lib.ptr(bpd), _lib.ptr(g2d), _lib.ptr(e2d), _lib.ptr(b1d), _lib.ptr(b2d),
                                    _lib.ptr(x3), rows, d, _lib.DTYPES[dt], None, None, 1e-6, None, _lib.ptr(ctr), _lib.stream_ptr()))
    assert torch.equal(x3, xd)
    assert lib.vittf_block_tail(_lib.ptr(ad), _lib.ptr(wpk), _lib.ptr(bpd), _lib.ptr(g2d), _lib.ptr(e2d), _lib.ptr(b1d), _lib.ptr(b2d),
                                _lib.ptr(xd), rows, 768, _lib.DTYPES[dt], None, None, 1e-6, None, _lib.ptr(ctr), _lib.stream_ptr()) == -1


def test_block_tail_two_streams_caller_owned_counters(gpu):
    """The library keeps no state of its own for the block tail: two calls in flight at once on two streams, each with its OWN
    tile counter (caller memory, vittf_block_tail_workspace_bytes()), write the same bits as the same calls one after the
    other; a missing or misaligned counter is refused."""
    lib = _lib.load()
    d, dt = 384, 'fp16'
    rows = (128 * 700 + 9, 128 * 650 + 100)                      # ~2.7 rounds of persistent workgroups each: they do overlap
    g = gen(77)
    wp = (torch.randn(d, d, generator=g) / d ** 0.5).to(TDT[dt]).to(gpu)
    w1 = (torch.randn(4 * d, d, generator=g) / d ** 0.5).to(TDT[dt]).to(gpu)
    w2 = (torch.randn(d, 4 * d, generator=g) / (4 * d) ** 0.5).to(TDT[dt]).to(gpu)
    wpk = vt.weights.pack_block_tail_weights(wp[None], w1[None], w2[None])[0].contiguous()
    vecs = [(0.3 * torch.randn(n, generator=g)).to(gpu) for n in (d, d, d, 4 * d, d, d, d)]      # bp, g2, e2, b1, b2, g1, e1
    vecs[1] += 1.0; vecs[5] += 1.0
    bp, g2, e2, b1, b2, g1, e1 = vecs
    assert lib.vittf_block_tail_workspace_bytes() == 4
    a = [torch.randn(r, d, generator=g).to(TDT[dt]).to(gpu) for r in rows]
    x0 = [(3 * torch.randn(r, d, generator=g)).to(gpu) for r in rows]

    def call(i, x, h, ctr, stream):
        return lib.vittf_block_tail(_lib.ptr(a[i]), _lib.ptr(wpk), _lib.ptr(bp), _lib.ptr(g2), _lib.ptr(e2), _lib.ptr(b1), _lib.ptr(b2),
                                    _lib.ptr(x), rows[i], d, _lib.DTYPES[dt], _lib.ptr(g1), _lib.ptr(e1), 1e-6, _lib.ptr(h),
                                    ctr, C.c_void_p(stream.cuda_stream))
    ctrs = torch.zeros(2, 64, dtype=torch.int32, device=gpu)      # two counters, 256 bytes apart
    cur = torch.cuda.current_stream()
    seq = []
    for i in range(2):
        x, h = x0[i].clone(), torch.empty(rows[i], d, dtype=TDT[dt], device=gpu)
        _lib.check(call(i, x, h, _lib.ptr(ctrs[i]), cur))
        seq.append((x, h))
    torch.cuda.synchronize()
    streams = [torch.cuda.Stream(), torch.cuda.Stream()]
    par = [(x0[i].clone(), torch.empty(rows[i], d, dtype=TDT[dt], device=gpu)) for i in range(2)]
    torch.cuda.synchronize()
    for rep in range(3):                                          # back to back on each stream: counters are re-zeroed per call
        for i in range(2):
            if rep:
                with torch.cuda.stream(streams[i]):
                    par[i][0].copy_(x0[i])
            _lib.check(call(i, par[i][0], par[i][1], _lib.ptr(ctrs[i]), streams[i]))
    torch.cuda.synchronize()
    for i in range(2):
        assert torch.equal(par[i][0], seq[i][0]) and torch.equal(par[i][1], seq[i][1]), f'stream {i}'
    assert call(0, par[0][0], par[0][1], None, cur) == -1
    assert call(0, par[0][0], par[0][1], C.c_void_p(ctrs.data_ptr() + 2), cur) == -1


# ------------------------------------------------------------------------------------------ attention
QSCALE = 0.125 * 1.4426950408889634      # log2(e) / 8, what VITTF_EPI_BIAS_QKV folds into q


def _prescale(qkv, heads, dt):
    """q third multiplied by log2(e)/8 in fp32 and rounded once, as the engine's qkv epilogue does."""
    out = qkv.float().clone()
    out[:, :heads * 64] *= QSCALE
    return out.to(TDT[dt])


def _attn_ref(qkv, batch, tokens, heads, pre=0):
    d = heads * 64
    x = qkv.double().view(batch, tokens, 3, heads, 64).permute(2, 0, 3, 1, 4)
    q, k, v = x[0], x[1], x[2]
    att = (q @ k.transpose(-2, -1)) * (np.log(2.0) if pre else 0.125)       # pre-scaled q: scores are in exp2 units
    return (att.softmax(-1) @ v).transpose(1, 2).reshape(batch * tokens, d)


# q as produced (the online-maximum kernel) | pre-scaled q: two 32-row blocks per wave taking turns (what the engine runs)
ATTN_VARIANTS = ['plain', 'pingpong']


def _attn_variant(variant):
    """-> the q_prescaled flag of vittf_attention, which selects the kernel."""
    return 0 if variant == 'plain' else 1


def _run_attn(gpu, qkv, batch, tokens, heads, dt, pre=0, pad_rows=2):
    lib = _lib.load()
    qd = qkv.to(gpu)
    out = torch.full((batch * tokens + pad_rows, heads * 64), 7.0, dtype=TDT[dt], device=gpu)
    _lib.check(lib.vittf_attention(_lib.ptr(qd), _lib.ptr(out), batch, tokens, heads, _lib.DTYPES[dt], pre, _lib.stream_ptr()))
    torch.cuda.synchronize()
    got = out.float().cpu().double()
    assert (got[batch * tokens:] == 7.0).all(), 'wrote past the last row'
    return got[:batch * tokens]


@pytest.mark.parametrize('variant', ATTN_VARIANTS)
@pytest.mark.parametrize('dt', ['bf16', 'fp16'])
@pytest.mark.parametrize('batch,tokens,heads', [(1, 1, 2), (2, 17, 2), (1, 33, 2), (1, 64, 2), (3, 65, 2), (1, 128, 6), (2, 129, 2),
                                                (1, 192, 2), (1, 200, 6), (2, 257, 2), (1, 320, 2), (2, 577, 2), (1, 1025, 2)])
def test_attention_small(gpu, dt, batch, tokens, heads, variant):
    """Token counts cover 1..17 key tiles of 64: every remainder path of the pipelined kernel's 3-deep ring."""
    pre = _attn_variant(variant)
    g = gen(batch * 1000 + tokens)
    qkv = torch.randn(batch * tokens, 3 * heads * 64, generator=g)
    qkv[:, :2 * heads * 64] *= 1.6          # logits with a std of ~2.5: a peaked softmax
    qkv = _prescale(qkv, heads, dt) if pre else qkv.to(TDT[dt])
    ref = _attn_ref(qkv, batch, tokens, heads, pre)
    got = _run_attn(gpu, qkv, batch, tokens, heads, dt, pre)
    vmax = float(qkv[:, 2 * heads * 64:].float().abs().max())
    # P is rounded to the 16-bit type before the second product, the output once more
    assert rel_fro(got, ref) <= 3 * EPS[dt]
    assert max_abs(got, ref) <= 6 * EPS[dt] * vmax


@pytest.mark.parametrize('variant', ATTN_VARIANTS)
@pytest.mark.parametrize('dt', ['bf16', 'fp16'])
@pytest.mark.parametrize('gain', [12.0, 60.0])
def test_attention_rescale_branch(gpu, dt, variant, gain):
    """Force the maximum to jump in a late key tile and in the ragged last tile: the online-softmax rescale path of
    the plain kernel and the overflow-triggered slow path of the lazy-maximum kernel (gain 60: scores 2^170 above
    the first tile's, past fp32's exponent range); a uniform-random check exercises neither."""
    pre = _attn_variant(variant)
    batch, tokens, heads = 1, 333, 2
    g = gen(4)
    qkv = torch.randn(batch * tokens, 3 * heads * 64, generator=g) * 0.5
    q = qkv[:, :128].view(tokens, 2, 64)
    k = qkv[:, 128:256].view(tokens, 2, 64)
    for key_row in (5, 100, 200, 332):                  # first tile, both halves of middle tiles, the ragged last tile
        k[key_row, 0] = q[7 + key_row % 50, 0] * gain   # one query row suddenly matches this key strongly
    k[40, 1] = q[3, 1] * -gain                          # and one strongly negative score
    qkv = _prescale(qkv, heads, dt) if pre else qkv.to(TDT[dt])
    ref = _attn_ref(qkv, batch, tokens, heads, pre)
    got = _run_attn(gpu, qkv, batch, tokens, heads, dt, pre)
    assert torch.isfinite(got).all()
    assert rel_fro(got, ref) <= 3 * EPS[dt]
    assert max_abs(got, ref) <= 6 * EPS[dt] * float(qkv[:, 256:].float().abs().max())


@pytest.mark.parametrize('variant', ATTN_VARIANTS)
@pytest.mark.parametrize('dt', ['bf16', 'fp16'])
def test_attention_full_size(gpu, dt, variant):
    """N = 4097 (512^2 image, P = 8), 6 heads: the headline shape; reference in fp64 on the GPU via torch ops."""
    pre = _attn_variant(variant)
    batch, tokens, heads = 2, 4097, 6
    g = gen(77)
    qkv = torch.randn(batch * tokens, 3 * heads * 64, generator=g)
    qkv[:, :2 * heads * 64] *= 1.5
    qkv = _prescale(qkv, heads, dt) if pre else qkv.to(TDT[dt])
    lib = _lib.load()
    qd = qkv.to(gpu)
    out = torch.zeros(batch * tokens, heads * 64, dtype=TDT[dt], device=gpu)
    try:
        _lib.check(lib.vittf_attention(_lib.ptr(qd), _lib.ptr(out), batch, tokens, heads, _lib.DTYPES[dt], pre, _lib.stream_ptr()))
        torch.cuda.synchronize()
    finally:
        _attn_variant('plain')
    x = qd.double().view(batch, tokens, 3, heads, 64)
    err2, ref2, mx = 0.0, 0.0, 0.0
    for b in range(batch):
        for h in range(heads):
            q, k, v = x[b, :, 0, h], x[b, :, 1, h], x[b, :, 2, h]
            r = ((q @ k.t()) * (np.log(2.0) if pre else 0.125)).softmax(-1) @ v
            d = out[b * tokens:(b + 1) * tokens, h * 64:(h + 1) * 64].double() - r
            err2 += float((d * d).sum()); ref2 += float((r * r).sum()); mx = max(mx, float(d.abs().max()))
    assert (err2 / ref2) ** 0.5 <= 3 * EPS[dt]
    assert mx <= 6 * EPS[dt] * float(qkv[:, 2 * heads * 64:].float().abs().max())


@pytest.mark.parametrize('variant', ATTN_VARIANTS)
@pytest.mark.parametrize('dt', ['bf16', 'fp16'])
@pytest.mark.parametrize('tokens', [65, 4097])
def test_attention_ignores_what_lies_behind_the_slices(gpu, dt, variant, tokens):
    """The ragged last key tile reaches past the last token of a slice.  Whatever lies there -- the next slice's rows, or
    for the last slice of a batch the engine's un-stored workspace padding, which is torch.empty memory -- must not reach
    the output: P = 0 times a NaN / Inf bit pattern would poison every row of the (slice, head).  The qkv buffer gets 80
    rows of 0x7FFF (NaN in both 16-bit types) behind the batch; results must be finite and equal to those of a buffer
    padded with zeros, bit for bit."""
    pre = _attn_variant(variant)
    batch, heads = 2, 2
    g = gen(tokens)
    qkv = torch.randn(batch * tokens, 3 * heads * 64, generator=g)
    qkv[:, :2 * heads * 64] *= 1.5
    qkv = _prescale(qkv, heads, dt) if pre else qkv.to(TDT[dt])
    pad = 80
    poisoned = torch.cat([qkv, torch.full((pad, qkv.shape[1]), float('nan')).to(TDT[dt])])
    assert bool(torch.isnan(poisoned[-pad:].float()).all())               # NaN bit patterns in the 16-bit type
    clean = torch.cat([qkv, torch.zeros(pad, qkv.shape[1]).to(TDT[dt])])
    got_p = _run_attn(gpu, poisoned, batch, tokens, heads, dt, pre)
    _attn_variant(variant)
    got_c = _run_attn(gpu, clean, batch, tokens, heads, dt, pre)
    assert torch.isfinite(got_p).all()
    assert torch.equal(got_p, got_c)
    if tokens <= 128:
        assert rel_fro(got_p, _attn_ref(qkv, batch, tokens, heads, pre)) <= 3 * EPS[dt]


# ------------------------------------------------------------------------------------------ front end
def test_volume_minmax(gpu):
    lib = _lib.load()
    for n in (1, 7, 1000, 64 * 64 * 64 + 3):
        v = torch.randn(n, generator=gen(n)) * 100
        vd = v.to(gpu)
        out = torch.zeros(2, device=gpu)
        ws = torch.empty(lib.vittf_minmax_workspace_bytes(), dtype=torch.uint8, device=gpu)
        _lib.check(lib.vittf_volume_minmax(_lib.ptr(vd), n, _lib.ptr(out), _lib.ptr(ws), ws.numel(), _lib.stream_ptr()))
        assert out.cpu().tolist() == [float(v.min()), float(v.max())]


@pytest.mark.parametrize('axis', ['z', 'y', 'x'])
@pytest.mark.parametrize('shape,im_sz', [((24, 16, 32), (24, 16, 32)), ((20, 12, 28), (16, 8, 16)), ((10, 10, 10), (32, 32, 32))])
def test_patch_embed_matches_prepare_tokens(gpu, axis, shape, im_sz):
    """gather + min-max + ImageNet normalise + nearest resize + conv + CLS + pos-embed, in exact fp32."""
    sd = vt.synthetic_state_dict(TINY_ARCH, 3)
    oracle = dino_vit.build_vit(TINY_ARCH, sd)
    vol = (torch.rand(shape, generator=gen(9)) * 300 - 100).half().float()
    imgs = ofv.normalized_slices(vol, axis)
    rows, cols = ofv.axis_image_size(im_sz, axis)
    with torch.no_grad():
        ref = oracle.prepare_tokens(F.interpolate(imgs, size=(rows, cols), mode='nearest'))
    model = vt.HipViT(sd, TINY_ARCH, 'fp16')
    dvol = vt.DeviceVolume(vol, gpu)
    view = dvol.view(axis, im_sz)
    pos, _, _ = model.pos_for(rows, cols)
    n_slices = imgs.shape[0]
    out = torch.zeros(n_slices, ref.shape[1], 128, device=gpu)
    _lib.check(model.lib.vittf_patch_embed(C.byref(model.cfg), C.byref(model.weights), C.byref(pos), C.byref(view), 0,
                                           n_slices, _lib.ptr(out), _lib.stream_ptr()))
    got = out.cpu()
    assert got.shape == ref.shape
    assert torch.allclose(got, ref, rtol=2e-5, atol=2e-5 * float(ref.abs().max()))


@pytest.mark.parametrize('axis', ['z', 'y', 'x'])
@pytest.mark.parametrize('shape,im_sz', [((24, 16, 32), (24, 16, 32)), ((20, 12, 28), (16, 8, 16)), ((10, 10, 10), (32, 32, 32)),
                                         ((40, 136, 136), (136, 136, 136))])       # 290 tokens x 40 slices: 45 full tiles + a ragged one
def test_patch_embed_vits8_on_the_matrix_cores(gpu, axis, shape, im_sz):
    """The ViT-S/8 form of the same step (D = 384, P = 8: patch_embed_mfma_kernel, fp16 head + tail operands on the matrix
    cores): against the oracle's prepare_tokens at the generic kernel's tolerance, and against the fp64 sum over the same fp32
    pixels and weights at 2e-6 of a row's largest value (the split drops lo x lo and rounds each tail to fp16: 2^-21)."""
    arch = (384, 1, 6, 8)
    sd = vt.synthetic_state_dict(arch, 3)
    oracle = dino_vit.build_vit(arch, sd)
    vol = (torch.rand(shape, generator=gen(11)) * 300 - 100).half().float()
    imgs = ofv.normalized_slices(vol, axis)
    rows, cols = ofv.axis_image_size(im_sz, axis)
    with torch.no_grad():
        x_in = F.interpolate(imgs, size=(rows, cols), mode='nearest')
        ref = oracle.prepare_tokens(x_in)
        ref64 = oracle.double().prepare_tokens(x_in.double())
    model = vt.HipViT(sd, arch, 'fp16')
    dvol = vt.DeviceVolume(vol, gpu)
    view = dvol.view(axis, im_sz)
    pos, _, _ = model.pos_for(rows, cols)
    n_slices = imgs.shape[0]
    out = torch.full((n_slices + 1, ref.shape[1], 384), 7.0, device=gpu)
    _lib.check(model.lib.vittf_patch_embed(C.byref(model.cfg), C.byref(model.weights), C.byref(pos), C.byref(view), 0,
                                           n_slices, _lib.ptr(out), _lib.stream_ptr()))
    got = out.cpu()
    assert (got[n_slices:] == 7.0).all(), 'wrote past the last row'
    got = got[:n_slices]
    assert got.shape == ref.shape
    assert torch.allclose(got, ref, rtol=2e-5, atol=2e-5 * float(ref.abs().max()))
    err = (got.double() - ref64).abs().amax(dim=-1)
    assert bool((err <= 2e-6 * ref64.abs().amax(dim=-1).clamp_min(1.0)).all()), float(err.max())
    # a second call over a sub-range of the slices writes the same bits (rows are independent of the tiling)
    if n_slices > 3:
        part = torch.zeros(2, ref.shape[1], 384, device=gpu)
        _lib.check(model.lib.vittf_patch_embed(C.byref(model.cfg), C.byref(model.weights), C.byref(pos), C.byref(view), 1, 2,
                                               _lib.ptr(part), _lib.stream_ptr()))
        assert torch.equal(part.cpu(), got[1:3])


# ------------------------------------------------------------------------------------------ pooling / axis sum
@pytest.mark.parametrize('axis', ['z', 'y', 'x'])
@pytest.mark.parametrize('S,n_out,f0,f1', [(32, 4, 3, 2), (10, 4, 4, 4), (7, 7, 2, 70), (130, 130, 1, 3), (16, 1, 5, 5)])
def test_pool_slices_bit_exact(gpu, axis, S, n_out, f0, f1):
    lib = _lib.load()
    d = 128
    k = (torch.randn(S, f0, f1, d, generator=gen(S + f1)) * 4).half()
    sl, (a, b) = vt.AXIS_DIMS[axis]
    order = [None] * 3
    order[sl], order[a], order[b] = 0, 1, 2
    full = k.permute(3, *order).contiguous()                      # reference layout (D, ...S at the axis...)
    size = list(full.shape[1:])
    size[sl] = n_out
    ref = F.adaptive_avg_pool3d(full, tuple(size))                # fp16 in / out, like infer.py:203,329
    n = [0, 0, 0]
    n[sl], n[a], n[b] = n_out, f0, f1
    shape, strides = vt.extract._slab_shape_strides(axis, d, n, n_out)
    dst = torch.zeros(shape, dtype=torch.float16, device=gpu)
    kd = k.to(gpu)
    _lib.check(lib.vittf_pool_slices(_lib.ptr(kd), 0, S, S, n_out, 0, n_out, f0, f1, d, _lib.ptr(dst), *strides,
                                     _lib.stream_ptr()))
    assert torch.equal(dst.cpu(), ref)
    # a sub-range of windows from a sub-range of resident slices gives the same bits (what a rank computes)
    if n_out >= 2:
        w0, nw = n_out // 2, n_out - n_out // 2
        s0 = vt.extract.window_bounds(w0, S, n_out)[0]
        shape2, strides2 = vt.extract._slab_shape_strides(axis, d, n, nw)
        dst2 = torch.zeros(shape2, dtype=torch.float16, device=gpu)
        ksub = k[s0:].contiguous().to(gpu)
        _lib.check(lib.vittf_pool_slices(_lib.ptr(ksub), s0, S - s0, S, n_out, w0, nw, f0, f1, d, _lib.ptr(dst2),
                                         *strides2, _lib.stream_ptr()))
        assert torch.equal(dst2.cpu(), ref.narrow(1 + sl, w0, nw))
    # windows whose slices are not resident are refused
    if S > 1:
        assert lib.vittf_pool_slices(_lib.ptr(kd), 1, S - 1, S, n_out, 0, n_out, f0, f1, d, _lib.ptr(dst), *strides,
                                     _lib.stream_ptr()) == -1


@pytest.mark.parametrize('world', [1, 2, 3])
@pytest.mark.parametrize('n', [(5, 4, 7), (6, 5, 16), (8, 8, 64), (3, 7, 24)])   # the last dims 16 / 64: the 8-values-per-thread kernel when the z chunk allows
def test_assemble_sum_bit_exact(gpu, world, n):
    lib = _lib.load()
    d = 128
    g = gen(world)
    vols = {ax: (torch.randn(d, *n, generator=g) * 8).half() for ax in 'zyx'}
    ref = ((0.0 + vols['z']) + vols['y']) + vols['x']            # fp16 adds, z -> y -> x (infer.py:330-332)
    chunks, gathered = [0, 0, 0], {}
    for ax in 'zyx':
        sl = vt.AXIS_DIMS[ax][0]
        c = -(-n[sl] // world)
        chunks[sl] = c
        pad = list(vols[ax].shape)
        pad[1 + sl] = c * world
        padded = torch.zeros(pad, dtype=torch.float16)
        padded.narrow(1 + sl, 0, n[sl]).copy_(vols[ax])
        gathered[ax] = torch.stack(padded.split(c, dim=1 + sl)).contiguous().to(gpu)     # [world, D, ...chunk...]
    out = torch.zeros(d, *n, dtype=torch.float16, device=gpu)
    carr = (C.c_int32 * 3)(*chunks)
    _lib.check(lib.vittf_assemble_sum(_lib.ptr(gathered['z']), _lib.ptr(gathered['y']), _lib.ptr(gathered['x']), world,
                                      carr, d, n[0], n[1], n[2], _lib.ptr(out), _lib.stream_ptr()))
    assert torch.equal(out.cpu(), ref)


# ------------------------------------------------------------------------------------------ similarity side
def test_sample_features_golden(gpu, golden_dir):
    g = load_golden(golden_dir, 'sampling.npz')
    feat, rel = torch.from_numpy(g['feat']), torch.from_numpy(g['rel'])
    for src in (feat, feat.half()):                               # fp32 and fp16 storage (values are fp16-exact)
        for mode in ('nearest', 'bilinear'):
            got = vt.sample_features3d(src.to(gpu), rel.clone(), mode)
            assert got.shape == (1, 1, 16, 32)
            ref = torch.from_numpy(g[mode])
            # same corner order and weight products as the CPU op; last-bit differences only
            assert torch.allclose(got[0, 0].cpu(), ref, rtol=0, atol=1e-6)


@pytest.fixture(params=['1', '0'])
def sim_split(request, monkeypatch):
    """both few-query similarity kernels: feature axis split over the workgroup's waves (default) / one thread per voxel pair"""
    monkeypatch.setenv('VITTF_SIM_SPLIT', request.param)
    return request.param


def test_similarity_golden(gpu, golden_dir, sim_split):
    g = load_golden(golden_dir, 'similarity.npz')
    feat = torch.from_numpy(g['feat'])
    vol = np.zeros(tuple(int(x) for x in g['vol_shape']), dtype=np.float32)
    ann = {'ntf1': torch.from_numpy(g['ann_ntf1']), 'ntf2': torch.from_numpy(g['ann_ntf2'])}
    got = vt.compute_similarities(vol, feat, ann)
    for k in ann:
        assert got[k].dtype == torch.uint8 and tuple(got[k].shape) == g[f'sim_{k}'].shape
        assert np.array_equal(got[k].numpy(), g[f'sim_{k}']), f'{k}: {(got[k].numpy() != g[f"sim_{k}"]).sum()} voxels differ'
    assert np.array_equal(vt.assign_labels(got), g['labels'])
    big = vt.compute_similarities(vol, feat, {'ntf1': torch.from_numpy(g['ann_big'])})
    diff = big['ntf1'].numpy().astype(int) - g['sim_big'].astype(int)
    # 1030 fp32 dot products summed in a different order than torch's einsum: allow isolated 1-LSB flips
    assert (diff != 0).sum() <= 2 and np.abs(diff[np.abs(diff) < 128]).max(initial=0) <= 1


def test_similarity_many_classes_and_chunks(gpu, sim_split):
    """17 + 16 + 1 + 40 annotations in 4 classes: chunks of 16 straddle class boundaries; vs the oracle."""
    g = gen(21)
    feat = F.normalize(torch.randn(64, 6, 7, 8, generator=g), dim=0)
    feat = F.normalize(feat + 0.7 * feat[:, 1:2, 2:3, 3:4], dim=0).half().float()
    shape = (12, 14, 16)
    ann = {f'c{i}': torch.stack([torch.randint(0, s, (n,), generator=g) for s in shape], 1)
           for i, n in enumerate((17, 16, 1, 40))}
    got = vt.compute_similarities(np.zeros(shape, np.float32), feat, ann)
    ref = osim.similarity_maps(shape, feat, ann)
    for k in ann:
        diff = got[k].numpy().astype(int) - ref[k].numpy().astype(int)
        assert (diff != 0).mean() <= 0.01, k
    labels = vt.assign_labels(got)
    assert labels.dtype == np.uint8 and labels.shape == tuple(s // 2 for s in shape)
    assert np.array_equal(labels, osim.assign_labels([got[k] for k in ann]))     # bit-exact on the same maps


def test_similarity_split_kernel_matches_plain_kernel(gpu, monkeypatch):
    """fp32 class maps of the two few-query kernels on a 32^3 x 384 volume, 3 classes / 21 annotations (two chunks):
    same values up to the summation order of the dot products."""
    lib = _lib.load()
    g = gen(77)
    f, n = 384, 32
    feat = F.normalize(torch.randn(f, n, n, n, generator=g), dim=0).half().to(gpu)
    qf = (feat.float().reshape(f, -1)[:, torch.randint(0, n ** 3, (21,), generator=g).to(gpu)].T
          + 0.02 * torch.randn(21, f, generator=g).to(gpu)).contiguous()      # queries near voxels of the volume
    starts = np.array([0, 5, 6, 21], np.int32)
    ws_bytes = lib.vittf_similarity_workspace_bytes(3, 0, 21)
    ws = torch.empty(ws_bytes, dtype=torch.uint8, device=gpu)
    maps = {}
    monkeypatch.setenv('VITTF_SIM_MFMA_MIN', '1000000')       # (21 annotations would take the matrix-core path)
    for split in ('1', '0'):
        monkeypatch.setenv('VITTF_SIM_SPLIT', split)
        out = torch.full((3 * n ** 3 + 1,), 7.0, device=gpu)
        _lib.check(lib.vittf_similarity_maps_f32(_lib.ptr(feat), 1, f, n, n, n, _lib.ptr(qf), starts.ctypes.data_as(C.POINTER(C.c_int32)),
                                                 3, 0, 0.0, None, _lib.ptr(out), _lib.ptr(ws), ws_bytes, _lib.stream_ptr()))
        assert float(out[-1]) == 7.0, 'wrote past the last map'
        maps[split] = out[:-1].cpu().view(3, -1)
    assert float(maps['1'].max()) > 0.01
    assert torch.allclose(maps['1'], maps['0'], rtol=2e-5, atol=1e-7)
    dots = torch.einsum('fv,af->av', feat.float().cpu().reshape(f, -1).double(), qf.cpu().double())
    act = torch.where(dots >= 0.25, dots, torch.zeros_like(dots)) ** 2.5
    ref = torch.stack([act[a:b].mean(0) for a, b in zip(starts[:-1], starts[1:])])
    edge = ((dots - 0.25).abs() < 1e-5).any(0)                      # the threshold decides differently in fp32 / fp64 there
    assert torch.allclose(maps['1'].double()[:, ~edge], ref[:, ~edge], rtol=1e-4, atol=1e-7)


def test_labels_bit_exact_random(gpu):
    g = gen(8)
    sims = [torch.randint(0, 256, (9, 10, 11), generator=g, dtype=torch.uint8) for _ in range(5)]
    assert np.array_equal(vt.assign_labels(sims), osim.assign_labels(sims))
    sims7 = sims + sims[:2]                                        # more maps than thresholds: extra ones are ignored
    assert np.array_equal(vt.assign_labels(sims7), osim.assign_labels(sims7))


def test_cosine_similarity_option(gpu, sim_split):
    """normalize=True: per-voxel L2 normalisation of the volume (F.normalize(feat, dim=0)) folded into the sampling and
    similarity kernels through a norm array, against the oracle that normalises the volume explicitly."""
    g = gen(33)
    feat = (torch.randn(64, 6, 7, 8, generator=g) * 3 + 0.5 * torch.randn(64, 1, 1, 1, generator=g)).half()
    feat[:, 0, 0, 0] = 0                                          # a zero voxel: eps clamp of F.normalize
    shape = (12, 14, 16)
    ann = {'a': torch.tensor([[2, 3, 4], [7, 7, 7], [11, 13, 15]]), 'b': torch.tensor([[0, 0, 0], [5, 9, 2]])}
    fd = feat.to(gpu)
    norms = vt.similarity.voxel_norms(fd).cpu()
    assert torch.allclose(norms, feat.float().norm(dim=0).clamp_min(1e-12), rtol=1e-6)
    got = vt.compute_similarities(np.zeros(shape, np.float32), feat, ann, normalize=True)
    ref = osim.similarity_maps(shape, feat.float(), ann, normalize=True)
    raw = vt.compute_similarities(np.zeros(shape, np.float32), feat, ann)
    for k in ann:
        d = (got[k].int() - ref[k].int()).abs()
        d = torch.minimum(d, 256 - d)
        assert int(d.max()) <= 1 and float((d > 0).float().mean()) <= 0.01, k
        assert not torch.equal(got[k], raw[k])                   # and it is not the un-normalised map
    again = vt.compute_similarities(np.zeros(shape, np.float32), feat, ann, voxel_norm=norms)   # norms passed in
    assert all(torch.equal(again[k], got[k]) for k in ann)


def test_voxel_norm_kernels(gpu):
    """vittf_voxel_norm: the split kernel (F % 4 == 0, voxels % 4 == 0) and the plain one on ragged shapes, against
    F.normalize's denominator max(|x|_2, 1e-12)."""
    g = gen(5)
    for shape in ((384, 16, 16, 16), (64, 6, 7, 8), (10, 3, 5, 7), (6, 1, 1, 3)):
        feat = (torch.randn(shape, generator=g) * 2).half()
        feat[:, 0, 0, 0] = 0
        got = vt.similarity.voxel_norms(feat.to(gpu)).cpu()
        want = feat.double().norm(dim=0).clamp_min(1e-12)
        assert got.shape == want.shape and torch.allclose(got.double(), want, rtol=2e-6, atol=0), shape
        assert float(got[0, 0, 0]) == pytest.approx(1e-12)


# ---------------------------------------------------------------- bilateral solver
def _u8_close(a, b, max_frac=0.01):
    d = (a.int() - b.int()).abs()
    d = torch.minimum(d, 256 - d)
    return int(d.max()) <= 1 and float((d > 0).float().mean()) <= max_frac


def test_bilateral_e2e_golden(gpu, golden_dir):
    """compute_similarities(..., bilateral_solver=True) against the reference's own output (fp64 CG on the device vs
    SciPy: the maps may differ by one quantisation step on isolated voxels)."""
    g = load_golden(golden_dir, 'bilateral.npz')
    ann = {k: torch.from_numpy(g[f'e2e_ann_{k}']) for k in ('ntf1', 'ntf2')}
    got = vt.compute_similarities(g['e2e_volume'], torch.from_numpy(g['e2e_feat']), ann, bilateral_solver=True)
    for k in ann:
        ref = torch.from_numpy(g[f'e2e_sim_{k}'])
        assert got[k].shape == ref.shape and got[k].dtype == torch.uint8
        assert _u8_close(got[k], ref), k


@pytest.mark.parametrize('shape,sim_shape', [((40, 36, 44), (20, 18, 22)), ((30, 30, 30), (30, 30, 30)), ((17, 50, 23), (31, 25, 40)),
                                             ((200, 160, 224), (100, 80, 112))])
def test_bilateral_refine_vs_oracle(gpu, shape, sim_shape):
    """refine_similarity (resizes, uint8 reference, crop, Sobel confidence, grid, bistochastisation, PCG, slice) against
    the CPU restatement at sizes with several spatial bins per axis, fp32 output compared directly."""
    from oracle import bilateral as obil
    gq = gen(sum(shape))
    zz, yy, xx = torch.meshgrid(*[torch.linspace(-1, 1, n) for n in shape], indexing='ij')
    blob = torch.exp(-3 * ((zz - 0.1) ** 2 + (yy + 0.2) ** 2 + xx ** 2))
    volume = (blob * 800 - 300 + 15 * torch.randn(shape, generator=gq)).float()
    n = tuple(max(4, s // 3) for s in shape)
    sim = torch.nn.functional.interpolate(blob[None, None], n, mode='trilinear')[0, 0]
    sim = (sim + 0.2 * torch.rand(n, generator=gq)).clamp(0, 1.2).float().contiguous()
    info = {}
    got = vt.bilateral.refine_similarity(sim.to(gpu), volume.to(gpu), sim_shape, info=info).cpu()
    ref = obil.refine_similarity(sim, volume, sim_shape)
    assert info['vertices'] > 8 and info['voxels'] > 0
    err = (got - ref).abs()
    # fp32 rounding of the trilinear resize can move a voxel across a uint8 / luma-bin edge (another vertex): rare
    assert float((err > 1e-4).float().mean()) < 0.01, float(err.max())
    assert float(err.median()) < 1e-6
    assert _u8_close(vt.bilateral.quantize_u8(got.to(gpu)).cpu(), osim.quantize_u8(ref)[0])


def test_bilateral_nothing_above_threshold(gpu):
    sim = torch.full((8, 8, 8), 0.05)
    vol = torch.rand((16, 16, 16), generator=gen(1))
    info = {}
    got = vt.bilateral.refine_similarity(sim.to(gpu), vol.to(gpu), (8, 8, 8), info=info).cpu()
    assert info['vertices'] == 0 and torch.equal(got, sim)


def test_quantize_wrap_u8(gpu):
    sim = torch.randn(5000, generator=gen(2)).abs() * 0.3
    sim[7] = sim.max() * 0.999          # lands on 256.x -> wraps to 0/1 like the x86 cast
    sim[11] = -0.01                     # the solver may undershoot
    got = vt.bilateral.quantize_u8(sim.to(gpu)).cpu()
    assert torch.equal(got, osim.quantize_u8(sim)[0])


# ---------------------------------------------------------------- resample_topk / take_most_dissimilar
def test_topk_voxels(gpu):
    lib = _lib.load()
    g = gen(40)
    maps = torch.randn(5, 3000, generator=g)
    maps[1, 10] = maps[1, 2000] = maps[1].max() + 1.0            # a tie at the top
    maps[2] = 0.5                                                # all equal: the first K indices
    maps[3, 77] = float('inf'); maps[3, 5] = -float('inf')
    md = maps.to(gpu)
    for k in (1, 4, 9):
        idx = torch.empty((5, k), dtype=torch.int32, device=gpu)
        _lib.check(lib.vittf_topk_voxels(_lib.ptr(md), 5, 3000, k, _lib.ptr(idx), _lib.stream_ptr()))
        for i in range(5):
            kth = torch.topk(maps[i], k).values[-1]
            want = (maps[i] >= kth).nonzero()[:k, 0]
            assert torch.equal(idx[i].cpu().long(), want), (i, k)


def test_resample_topk_golden(gpu, golden_dir):
    g = load_golden(golden_dir, 'refinement.npz')
    feat, sims = torch.from_numpy(g['rt_feat']), torch.from_numpy(g['rt_sims'])
    import infer as drop_in
    for K, expo, mode in ((3, 2.0, 'nearest'), (8, 1.5, 'bilinear')):
        got = drop_in.resample_topk(feat, sims, K, expo, mode).cpu()
        ref = torch.from_numpy(g[f'rt_out_K{K}'])
        assert got.shape == ref.shape and got.dtype == torch.float32
        assert float((got - ref).abs().max()) < 2e-5
    half = drop_in.resample_topk(feat.half(), sims[0], 3, 2.0, 'nearest').cpu()       # fp16 volume, 5-D sims
    assert half.dtype == torch.float16 and half.shape == ref.shape
    assert float((half.float() - osim.resample_topk(feat.half().float(), sims, 3, 2.0, 'nearest')).abs().max()) < 2e-3


def test_take_most_dissimilar_golden(gpu, golden_dir):
    g = load_golden(golden_dir, 'refinement.npz')
    x = torch.from_numpy(g['md_x'])
    import infer as drop_in
    for measure in ('cosine', 'euclidean'):
        got = drop_in.take_most_dissimilar(x, 9, measure)
        assert sorted(map(tuple, got.tolist())) == sorted(map(tuple, g[f'md_{measure}'].tolist())), measure
        lib = _lib.load()
        xd = x.to(gpu)
        dist = torch.empty(x.shape[0], device=gpu)
        _lib.check(lib.vittf_mean_pairwise_distance(_lib.ptr(xd), x.shape[0], x.shape[1], 0 if measure == 'cosine' else 1,
                                                    _lib.ptr(dist), _lib.stream_ptr()))
        assert torch.allclose(dist.cpu(), osim.mean_pairwise_distance(x, measure), atol=2e-6, rtol=1e-5)
    assert drop_in.take_most_dissimilar(x[:5], 9) is x[:5] or torch.equal(drop_in.take_most_dissimilar(x[:5], 9), x[:5])
    with pytest.raises(ValueError):
        drop_in.take_most_dissimilar(x, 9, 'manhattan')


@pytest.mark.parametrize('normalize', [False, True])
def test_similarity_many_annotations_mfma_path(gpu, normalize):
    """A >= 64 and F = 384 take the matrix-core path (sim_mfma.hip: volume read once, fp16 hi + lo queries, classes
    padded to 32-query chunks): against the oracle, incl. a 1-annotation class and a voxel count that is not a multiple
    of the 256-voxel workgroup tile."""
    g = gen(77 + int(normalize))
    feat = torch.nn.functional.normalize(torch.randn(384, 20, 18, 22, generator=g), dim=0)
    feat = (feat + 0.7 * feat[:, 5:6, 6:7, 7:8]).half()
    if not normalize:
        feat = torch.nn.functional.normalize(feat.float(), dim=0).half()
    shape = (40, 36, 44)
    ann = {'a': torch.randint(0, 36, (100, 3), generator=g), 'b': torch.tensor([[10, 12, 14]]),
           'c': torch.randint(0, 36, (300, 3), generator=g)}
    got = vt.compute_similarities(np.zeros(shape, np.float32), feat, ann, normalize=normalize)
    ref = osim.similarity_maps(shape, feat.float(), ann, normalize=normalize)
    for k in ann:
        assert int(ref[k].max()) > 20
        d = (got[k].int() - ref[k].int()).abs()
        d = torch.minimum(d, 256 - d)
        assert int(d.max()) <= 1 and float((d > 0).float().mean()) <= 0.01, (k, int(d.max()), float((d > 0).float().mean()))


@pytest.mark.parametrize('grid,vol_shape', [((8, 8, 8), (64, 64, 64)), ((6, 7, 5), (60, 70, 64)), ((16, 16, 16), (34, 20, 96)),
                                             ((8, 8, 8), (16, 16, 30))])
def test_quantize_16_per_thread_matches_bytewise_kernel(gpu, monkeypatch, grid, vol_shape):
    """The uint8 maps (quantise + nearest resize to vol.shape // 2, predict_ntf.py:95-100) from the 16-values-per-thread kernel
    and from the one-byte-per-thread kernel are the same bytes: integer and non-integer resize ratios, two classes, wrapped
    values, and an output row length the vector kernel does not take (15: both runs use the bytewise kernel)."""
    g = gen(sum(grid) + sum(vol_shape))
    feat = torch.nn.functional.normalize(torch.randn(64, *grid, generator=g), dim=0).half()
    ann = {'a': torch.stack([torch.randint(0, s, (3,), generator=g) for s in vol_shape], 1),
           'b': torch.stack([torch.randint(0, s, (2,), generator=g) for s in vol_shape], 1)}
    vol = np.zeros(vol_shape, np.float32)
    monkeypatch.setenv('VITTF_SIM_QUANT16', '1')
    fast = vt.compute_similarities(vol, feat, ann)
    monkeypatch.setenv('VITTF_SIM_QUANT16', '0')
    slow = vt.compute_similarities(vol, feat, ann)
    ref = osim.similarity_maps(vol_shape, feat.float(), ann)
    for k in ann:
        assert tuple(fast[k].shape) == tuple(s // 2 for s in vol_shape) and int(fast[k].max()) > 0
        assert torch.equal(fast[k], slow[k]), k
        assert float((fast[k] != ref[k]).float().mean()) <= 0.01


@pytest.mark.parametrize('grid,vol_shape,counts', [((64, 64, 64), (512, 512, 512), (16,)),      # the bench's query: matrix cores, x 4 resize
                                                   ((16, 16, 16), (64, 64, 64), (3, 2)),            # VALU kernel, x 2
                                                   ((16, 16, 16), (32, 32, 32), (9,)),              # x 1
                                                   ((8, 8, 8), (16, 16, 256), (5, 40)),             # x 16 along the last dim
                                                   ((6, 7, 5), (60, 70, 64), (4, 1))])              # ratios 5, 5, 6.4: the general form
def test_similarity_query_one_call_same_bytes(gpu, monkeypatch, grid, vol_shape, counts):
    """vittf_similarity_query (what compute_similarities runs for up to 64 annotations: coordinates as a kernel argument, the
    maxima zeroed by the sampling kernel, the power-of-two form of the quantise kernel) against the three separate steps it
    replaces -- coordinates copied to the device, vittf_sample_features, vittf_similarity -- with the general 16-value quantise
    kernel and with the bytewise one: the same bytes, from a workspace that starts as 0xff."""
    lib = _lib.load()
    g = gen(sum(grid) + sum(counts))
    f = 384
    feat = torch.nn.functional.normalize(torch.randn(f, *grid, generator=g), dim=0)
    feat = torch.nn.functional.normalize((feat + 0.7 * feat[:, 2:3, 3:4, 4:5]).half().float(), dim=0).half().to(gpu)
    ann = {f'c{i}': torch.stack([torch.randint(0, s, (n,), generator=g) for s in vol_shape], 1) for i, n in enumerate(counts)}
    got = vt.compute_similarities(_ShapeOnly(vol_shape), feat, ann,
                                  keep_on_device=True)
    # the separate steps, through the C ABI
    coords = torch.cat([ann[k] for k in ann])
    rel = ((coords.float() + 0.5) / torch.tensor([list(vol_shape)], dtype=torch.float32) * 2.0 - 1.0).to(gpu).contiguous()
    a = rel.shape[0]
    qf = torch.empty(a, f, device=gpu)
    n0, n1, n2 = grid
    _lib.check(lib.vittf_sample_features(_lib.ptr(feat), 1, f, n0, n1, n2, _lib.ptr(rel), a, _lib.SAMPLE_MODES['bilinear'], None,
                                         _lib.ptr(qf), _lib.stream_ptr()))
    starts = np.concatenate(([0], np.cumsum(counts))).astype(np.int32)
    o = tuple(s // 2 for s in vol_shape)
    for q16 in ('2', '0'):
        monkeypatch.setenv('VITTF_SIM_QUANT16', q16)
        ws = torch.full((lib.vittf_similarity_workspace_bytes(len(counts), n0 * n1 * n2, a),), 0xff, dtype=torch.uint8, device=gpu)
        out = torch.empty((len(counts), *o), dtype=torch.uint8, device=gpu)
        _lib.check(lib.vittf_similarity(_lib.ptr(feat), f, n0, n1, n2, _lib.ptr(qf), starts.ctypes.data_as(C.POINTER(C.c_int32)),
                                        len(counts), 0, None, o[0], o[1], o[2], _lib.ptr(out), _lib.ptr(ws), ws.numel(),
                                        _lib.stream_ptr()))
        for i, k in enumerate(ann):
            assert int(out[i].max()) > 0
            assert torch.equal(got[k], out[i]), (k, q16)
    monkeypatch.setenv('VITTF_SIM_QUANT16', '1')
    # the entry itself from a poisoned workspace; more than VITTF_QUERY_MAX_A annotations are refused
    ws = torch.full((lib.vittf_similarity_query_workspace_bytes(len(counts), n0 * n1 * n2, a, f),), 0xff, dtype=torch.uint8, device=gpu)
    out2 = torch.empty((len(counts), *o), dtype=torch.uint8, device=gpu)
    rel_h = np.ascontiguousarray(rel.cpu().numpy())
    _lib.check(lib.vittf_similarity_query(_lib.ptr(feat), f, n0, n1, n2, rel_h.ctypes.data_as(C.POINTER(C.c_float)),
                                          starts.ctypes.data_as(C.POINTER(C.c_int32)), len(counts), 0, None, o[0], o[1], o[2],
                                          _lib.ptr(out2), _lib.ptr(ws), ws.numel(), _lib.stream_ptr()))
    assert torch.equal(out2, out)
    too_many = np.array([0, _lib.QUERY_MAX_A + 1], np.int32)
    assert lib.vittf_similarity_query(_lib.ptr(feat), f, n0, n1, n2, rel_h.ctypes.data_as(C.POINTER(C.c_float)),
                                      too_many.ctypes.data_as(C.POINTER(C.c_int32)), 1, 0, None, o[0], o[1], o[2], _lib.ptr(out2),
                                      _lib.ptr(ws), ws.numel(), _lib.stream_ptr()) == -1


class _ShapeOnly:
    """compute_similarities only reads volume.shape unless the bilateral solver is asked for."""
    def __init__(self, shape):
        self.shape = tuple(shape)


@pytest.mark.parametrize('grid,counts,min_a', [((7, 9, 11), (70, 3, 33), None),            # 693 voxels: rows not 16-byte aligned -> strided loads
                                                ((16, 16, 16), (2,) * 40, None),              # 40 classes: tables through device memory
                                                ((32, 32, 32), (16,), 8), ((24, 20, 18), (5, 1, 9), 8),   # few queries on the matrix cores
                                                ((24, 20, 18), (30,), None),                  # one class, one chunk: the persistent kernel, partial last tile
                                                ((48, 48, 50), (20,), None), ((64, 64, 64), (16,), 8),   # ... with 1-2 / 4 tiles per workgroup
                                                ((64, 8, 8), (1024, 1024), None)])
def test_similarity_mfma_path_shapes(gpu, monkeypatch, grid, counts, min_a):
    """The matrix-core similarity path on the shapes around its fast case: voxel counts whose feature rows cannot be
    fetched as aligned 16-byte chunks, more classes than its by-value class table holds, few queries (threshold lowered
    with VITTF_SIM_MFMA_MIN), two full 1024-annotation classes -- each against the oracle and against the VALU kernels."""
    g = gen(sum(grid) + len(counts))
    feat = torch.nn.functional.normalize(torch.randn(384, *grid, generator=g), dim=0)
    feat = torch.nn.functional.normalize((feat + 0.7 * feat[:, 2:3, 3:4, 4:5]).half().float(), dim=0).half()
    shape = tuple(2 * s for s in grid)
    ann = {f'c{i}': torch.stack([torch.randint(0, s, (n,), generator=g) for s in shape], 1) for i, n in enumerate(counts)}
    if min_a is not None:
        monkeypatch.setenv('VITTF_SIM_MFMA_MIN', str(min_a))
    got = vt.compute_similarities(np.zeros(shape, np.float32), feat, ann)
    monkeypatch.setenv('VITTF_SIM_MFMA', '0')                                 # (read once per process in the library: see below)
    monkeypatch.setenv('VITTF_SIM_MFMA_MIN', '1000000')
    valu = vt.compute_similarities(np.zeros(shape, np.float32), feat, ann)
    ref = osim.similarity_maps(shape, feat.float(), ann)
    for k in ann:
        for name, other in (('oracle', ref[k]), ('VALU kernels', valu[k])):
            d = (got[k].int() - other.int()).abs()
            d = torch.minimum(d, 256 - d)
            assert int(d.max()) <= 1 and float((d > 0).float().mean()) <= 0.01, (k, name, int(d.max()), float((d > 0).float().mean()))


@pytest.mark.parametrize('grid,counts', [((24, 20, 18), (70, 3, 33)),          # 8640 voxels: a partial last 512-voxel workgroup
                                         ((16, 16, 16), (1,) * 5 + (40,)),       # 8 workgroups, one-annotation classes
                                         ((64, 8, 8), (1024, 1024))])
def test_similarity_mfma_two_voxel_blocks_per_wave(gpu, monkeypatch, grid, counts):
    """sim_mfma_kernel with two 32-voxel blocks per wave (512-voxel workgroups: the shape the 5 x 1024-query preset runs at
    size) forced on small volumes: every voxel goes through the same MFMA sequence as with one block per wave, so the maps
    are the same BYTES; and both match the oracle."""
    g = gen(sum(grid) + len(counts) + 5)
    feat = torch.nn.functional.normalize(torch.randn(384, *grid, generator=g), dim=0)
    feat = torch.nn.functional.normalize((feat + 0.7 * feat[:, 2:3, 3:4, 4:5]).half().float(), dim=0).half()
    shape = tuple(2 * s for s in grid)
    ann = {f'c{i}': torch.stack([torch.randint(0, s, (n,), generator=g) for s in shape], 1) for i, n in enumerate(counts)}
    monkeypatch.setenv('VITTF_SIM_MFMA_MIN', '2')
    monkeypatch.setenv('VITTF_SIM_MFMA_VB', '2')
    two = vt.compute_similarities(np.zeros(shape, np.float32), feat, ann)
    assert _lib.kernel_name('similarity') == 'sim_mfma_kernel<2 voxel blocks>'
    monkeypatch.setenv('VITTF_SIM_MFMA_VB', '1')
    one = vt.compute_similarities(np.zeros(shape, np.float32), feat, ann)
    assert _lib.kernel_name('similarity') == 'sim_mfma_kernel'
    ref = osim.similarity_maps(shape, feat.float(), ann)
    for k in ann:
        assert torch.equal(two[k], one[k]), k
        d = (two[k].int() - ref[k].int()).abs()
        d = torch.minimum(d, 256 - d)
        assert int(d.max()) <= 1 and float((d > 0).float().mean()) <= 0.01, (k, int(d.max()), float((d > 0).float().mean()))


@pytest.mark.parametrize('grid,counts', [((16, 16, 16), (70, 3, 33)), ((24, 20, 18), (1024, 200)), ((16, 16, 16), (2,) * 40)])
def test_similarity_mfma_768_features(gpu, monkeypatch, grid, counts):
    """ViT-B/8 feature volumes (F = 768) on the matrix-core similarity kernel: a 32-query chunk is two 384-feature units, the
    accumulators run over both -- against the oracle and against the VALU kernels (which took every F != 384 before)."""
    g = gen(sum(grid) + len(counts) + 768)
    feat = torch.nn.functional.normalize(torch.randn(768, *grid, generator=g), dim=0)
    feat = torch.nn.functional.normalize((feat + 0.7 * feat[:, 2:3, 3:4, 4:5]).half().float(), dim=0).half()
    shape = tuple(2 * s for s in grid)
    ann = {f'c{i}': torch.stack([torch.randint(0, s, (n,), generator=g) for s in shape], 1) for i, n in enumerate(counts)}
    got = vt.compute_similarities(np.zeros(shape, np.float32), feat, ann)
    assert _lib.kernel_name('similarity') == 'sim_mfma_kernel<F 768>'
    monkeypatch.setenv('VITTF_SIM_MFMA', '0')
    monkeypatch.setenv('VITTF_SIM_MFMA_MIN', '1000000')
    valu = vt.compute_similarities(np.zeros(shape, np.float32), feat, ann)
    assert _lib.kernel_name('similarity').startswith('sim_accumulate')
    ref = osim.similarity_maps(shape, feat.float(), ann)
    for k in ann:
        for name, other in (('oracle', ref[k]), ('VALU kernels', valu[k])):
            d = (got[k].int() - other.int()).abs()
            d = torch.minimum(d, 256 - d)
            assert int(d.max()) <= 1 and float((d > 0).float().mean()) <= 0.01, (k, name, int(d.max()), float((d > 0).float().mean()))


@pytest.mark.parametrize('dt', ['bf16', 'fp16'])
@pytest.mark.parametrize('rows,k', [(256, 384), (1030, 1536), (33000, 384), (4097 * 3, 1536), (100, 384)])
def test_gemm_residual_ln(gpu, dt, rows, k):
    """x += a . w^T + bias and h = LayerNorm(x) in one call (whole-row GEMM with the LayerNorm in its epilogue + the tiled /
    LayerNorm kernels for the rows beyond the last full 256-row tile) against fp64 and against the separate kernels."""
    _residual_ln_case(gpu, dt, rows, 384, k)


@pytest.mark.parametrize('dt', ['bf16', 'fp16'])
@pytest.mark.parametrize('rows,k', [(128, 768), (1030, 3072), (33000, 768), (4097 * 2, 3072), (100, 768), (257, 64)])
def test_gemm_residual_ln_768_columns(gpu, dt, rows, k):
    """The same with 768 output columns (ViT-B/8 proj and fc2: 8 waves side by side, 128-row workgroups)."""
    _residual_ln_case(gpu, dt, rows, 768, k)


def _residual_ln_case(gpu, dt, rows, n, k):
    lib = _lib.load()
    a, w, bias, ref = _gemm_inputs(rows, n, k, dt, rows + k)
    g = gen(rows)
    x0 = torch.randn(rows + 2, n, generator=g) * 3.0
    lg, lb = 1.0 + 0.2 * torch.randn(n, generator=g), 0.1 * torch.randn(n, generator=g)
    ad, wd, bd, gd, ld = a.to(gpu), w.to(gpu), bias.to(gpu), lg.to(gpu), lb.to(gpu)
    xd = x0.to(gpu)
    hd = torch.full((rows + 2, n), 7.0, dtype=TDT[dt], device=gpu)
    _lib.check(lib.vittf_gemm_residual_ln(_lib.ptr(ad), _lib.ptr(wd), _lib.ptr(bd), _lib.ptr(xd), rows, n, k, _lib.DTYPES[dt],
                                          _lib.ptr(gd), _lib.ptr(ld), 1e-6, _lib.ptr(hd), _lib.stream_ptr()))
    gx, gh = xd.cpu().double(), hd.float().cpu().double()
    assert torch.equal(gx[rows:], x0[rows:].double()) and (gh[rows:] == 7.0).all(), 'wrote past the last row'
    xref = x0[:rows].double() + ref
    assert ((gx[:rows] - xref).abs() <= 1e-5 * k ** 0.5 + 1e-6 * ref.abs()).all()
    href = F.layer_norm(gx[:rows], (n,), lg.double(), lb.double(), 1e-6)           # LayerNorm of the x the kernel produced
    assert ((gh[:rows] - href).abs() <= EPS[dt] * href.abs() * 1.01 + 2e-5).all()
    # the separate kernels give the same bits for x and h
    x2 = x0.to(gpu)
    h2 = torch.empty(rows, n, dtype=TDT[dt], device=gpu)
    _lib.check(lib.vittf_gemm(_lib.ptr(ad), _lib.ptr(wd), _lib.ptr(bd), _lib.ptr(x2), rows, n, k, _lib.EPI_BIAS_RESIDUAL, 0,
                              _lib.DTYPES[dt], _lib.stream_ptr()))
    _lib.check(lib.vittf_layernorm(_lib.ptr(x2), _lib.ptr(gd), _lib.ptr(ld), _lib.ptr(h2), rows, n, 1e-6, _lib.DTYPES[dt],
                                   _lib.stream_ptr()))
    assert torch.equal(x2.cpu(), xd.cpu())
    assert float((h2.float().cpu() - hd[:rows].float().cpu()).abs().max()) <= 2 * EPS[dt] * float(h2.float().abs().max())


@pytest.mark.parametrize('shape', [(32, 32, 32), (17, 70, 9), (5, 3, 130), (1, 1, 1), (3, 5, 16), (2, 70, 1024), (4, 1, 8)])
@pytest.mark.parametrize('connectivity', [1, 2, 3, 4])
def test_surface_shell_matches_scipy(gpu, shape, connectivity):
    """vittf_erode_mask / vittf_surface_shell against scipy.ndimage.binary_erosion (the reference's own dependency,
    compare_feat_sampling.py:19-24) bit for bit: blobs touching the border, ragged shapes, all structuring elements."""
    from scipy.ndimage import binary_erosion, generate_binary_structure
    from oracle import samplers as osmp
    lib = _lib.load()
    g = gen(sum(shape) + connectivity)
    lab = (torch.rand(shape, generator=g) < 0.85).to(torch.uint8) * 3          # class 3, dense enough to survive two erosions
    lab[torch.rand(shape, generator=g) < 0.03] = 1                              # another class sprinkled in
    d = lab.to(gpu)
    out = torch.full_like(d, 9)
    _lib.check(lib.vittf_erode_mask(_lib.ptr(d), *shape, 3, connectivity, _lib.ptr(out), _lib.stream_ptr()))
    ref = binary_erosion(lab.numpy() == 3, generate_binary_structure(3, connectivity))
    assert np.array_equal(out.cpu().numpy().astype(bool), ref) and int(out.max()) <= 1
    _lib.check(lib.vittf_erode_mask(_lib.ptr(d), *shape, -1, connectivity, _lib.ptr(out), _lib.stream_ptr()))
    assert np.array_equal(out.cpu().numpy().astype(bool), binary_erosion(lab.numpy() != 0, generate_binary_structure(3, connectivity)))
    shell = vt.samplers.surface_shell(d, connectivity, class_id=3)
    assert np.array_equal(shell.cpu().numpy().astype(bool), osmp.surface_shell(lab.numpy() == 3, connectivity))
    assert np.array_equal(vt.samplers.surface_shell(lab.numpy() == 3, connectivity).cpu().numpy().astype(bool),
                          osmp.surface_shell(lab.numpy() == 3, connectivity))


def test_surface_shell_golden(gpu, golden_dir):
    """vittf_surface_shell / samplers.sample_surface against the reference's own sample_surface output
    (tests/golden/samplers.npz: the whole shell in index order), both classes of one label upload, all elements."""
    g = load_golden(golden_dir, 'samplers.npz')
    dl = vt.samplers.device_labels(g['labels'])
    for cls in (2, 5):
        for dist in (1, 2, 3, 4):
            want = torch.from_numpy(g[f'shell_c{cls}_d{dist}'].astype(np.int64))
            assert torch.equal(vt.samplers.surface_shell(dl, dist, class_id=cls).nonzero().cpu(), want), (cls, dist)
            assert torch.equal(vt.samplers.sample_surface(dl, 10 ** 6, dist_from_surface=dist, class_id=cls), want)
            assert torch.equal(vt.samplers.sample_surface(g['labels'] == cls, 10 ** 6, dist_from_surface=dist), want)


def test_samplers_draw_from_the_reference_candidate_sets(gpu):
    """sample_uniform / sample_surface / sample_both (compare_feat_sampling.py:13-33): distinct voxels, all inside the
    class mask / the scipy shell; fewer shell voxels than requested -> the whole shell in index order, like the reference."""
    from oracle import samplers as osmp
    _, lab = vt.synthetic_volume('sphere_filled', 48)
    labels = (lab.to(torch.uint8) * 2)
    shell = osmp.surface_shell(lab.numpy())
    torch.manual_seed(0)
    dl = vt.samplers.device_labels(labels)
    u = vt.samplers.sample_uniform(dl, 50, thin_to_reasonable=True, class_id=2)
    assert u.shape == (50, 3) and u.dtype == torch.int64 and not u.is_cuda
    assert bool(lab[u[:, 0], u[:, 1], u[:, 2]].all()) and len({tuple(r) for r in u.tolist()}) == 50
    s = vt.samplers.sample_surface(lab.numpy(), 40)
    assert s.shape == (40, 3) and bool(shell[s[:, 0], s[:, 1], s[:, 2]].all()) and len({tuple(r) for r in s.tolist()}) == 40
    everything = vt.samplers.sample_surface(dl, 10 ** 6, class_id=2)
    assert torch.equal(everything, torch.as_tensor(shell).nonzero())
    b = vt.samplers.sample_both(dl, 30, class_id=2)
    assert b.shape == (30, 3) and bool(shell[b[15:, 0], b[15:, 1], b[15:, 2]].all())


@pytest.mark.parametrize('n,classes', [(0, 2), (1, 2), (1000, 2), (16 * 4096 + 7, 6), (3 * 10 ** 6 + 5, 16)])
def test_confusion_matrix_matches_numpy(gpu, n, classes):
    """vittf_confusion_matrix against np.add.at (what sklearn's confusion_matrix counts; predict_ntf.py:228-246): exact
    int64 counts, aligned and unaligned pointers, tails, out-of-range labels reported."""
    lib = _lib.load()
    g = gen(n + classes)
    t = torch.randint(0, classes, (n + 1,), generator=g, dtype=torch.uint8)
    p = torch.where(torch.rand(n + 1, generator=g) < 0.7, t, torch.randint(0, classes, (n + 1,), generator=g, dtype=torch.uint8))
    for off in (0, 1):                                     # off = 1: pointers not 16-byte aligned
        tt, pp = t[off:off + n], p[off:off + n]
        ref = np.zeros((classes, classes), np.int64)
        np.add.at(ref, (tt.numpy().astype(np.int64), pp.numpy().astype(np.int64)), 1)
        td, pd = t.to(gpu)[off:off + n], p.to(gpu)[off:off + n]
        counts = torch.full((classes * classes + 1,), -5, dtype=torch.int64, device=gpu)
        _lib.check(lib.vittf_confusion_matrix(_lib.ptr(td), _lib.ptr(pd), n, classes, _lib.ptr(counts), _lib.stream_ptr()))
        got = counts.cpu().numpy()
        assert got[-1] == 0 and np.array_equal(got[:-1].reshape(classes, classes), ref)
    if n >= 1000:
        ref0 = np.zeros((classes, classes), np.int64)
        np.add.at(ref0, (t[:n].numpy().astype(np.int64), p[:n].numpy().astype(np.int64)), 1)
        assert np.array_equal(vt.scores.confusion_matrix(t[:n], p[:n], classes), ref0)
        bad = t[:n].clone(); bad[n // 2] = classes
        with pytest.raises(ValueError):
            vt.scores.confusion_matrix(bad, p[:n], classes)
        # derived scores = oracle's restatement of the sklearn figures (binary case of evaluate_similarities.py:65-68)
        tb, pb = (t[:n] > 0).to(torch.uint8), (p[:n] > 1).to(torch.uint8)
        acc, prec, rec, f1, iou, cm = vt.scores.scores(tb, pb)
        want = osim.evaluate_predictions(pb.numpy(), tb.numpy())
        assert acc == want['accuracy'] and prec.tolist() == want['precision'] and rec.tolist() == want['recall']
        assert f1.tolist() == want['f1'] and iou.tolist() == want['iou'] and cm.tolist() == want['confusion_matrix']


# ---------------------------------------------------------------- volume I/O + resize formats
@pytest.mark.parametrize('shape,out', [((8, 9, 10), (16, 18, 20)), ((7, 5, 3), (13, 11, 17)), ((16, 16, 16), (8, 8, 8)),
                                       ((5, 6, 40), (10, 12, 33)), ((12, 10, 64), (24, 20, 128)), ((3, 3, 3), (3, 3, 3))])
def test_resize_nearest_u8_matches_interpolate(gpu, shape, out):
    """predict_ntf.py:217-218 (label up-sample) and evaluate_similarities.py:63 (class-mask resize): bit-exact against
    F.interpolate(mode='nearest') on the CPU, aligned (16-byte stores) and ragged fast dims, up- and down-sampling."""
    lab = torch.randint(0, 6, shape, generator=gen(sum(shape)), dtype=torch.uint8)
    ref = F.interpolate(lab[None, None], out, mode='nearest')[0, 0]
    got = vt.scores.resize_nearest_u8(lab.numpy(), out)
    assert got.dtype == np.uint8 and got.shape == out and np.array_equal(got, ref.numpy())
    for cls in (0, 3):
        mask_ref = F.interpolate((lab == cls).to(torch.uint8)[None, None], out, mode='nearest')[0, 0]
        got = vt.scores.resize_nearest_u8(lab, out, equals=cls, keep_on_device=True)
        assert got.is_cuda and torch.equal(got.cpu(), mask_ref)
    lib = _lib.load()
    d = lab.to(gpu)
    assert lib.vittf_resize_nearest_u8(_lib.ptr(d), 8, 9, 10, _lib.ptr(d), 8, 9, 10, -1, _lib.stream_ptr()) == -1   # in place


def test_fp16_volume_is_widened_on_the_device(gpu):
    """A volume stored as fp16 (create_synthetic_volumes.py:44-46) is uploaded as 2-byte values and widened in HBM: same
    fp32 values, min / max and features as the host-side .float() of infer.py:137."""
    for shape in ((24, 16, 32), (5, 7, 9)):
        vol = (torch.randn(shape, generator=gen(3)) * 300).half()
        a = vt.DeviceVolume(vol, gpu)
        b = vt.DeviceVolume(vol.float(), gpu)
        assert a.data.dtype == torch.float32 and torch.equal(a.data, b.data) and torch.equal(a.minmax, b.minmax)
        assert torch.equal(a.data.cpu(), vol.float())


def test_empty_class_keeps_its_key(gpu):
    """An annotation class without points keeps its key with an all-zero map (the reference: mean over an empty slice ->
    NaN -> uint8 0, predict_ntf.py:46-49, 70-72, 99), so the label ids of the later classes do not shift."""
    g = gen(5)
    feat = F.normalize(torch.randn(64, 6, 7, 8, generator=g), dim=0).half()
    shape = (12, 14, 16)
    full = {'a': torch.tensor([[2, 3, 4], [7, 7, 7]]), 'b': torch.tensor([[5, 9, 2], [1, 1, 1], [11, 13, 15]])}
    holed = {'a': full['a'], 'gap': torch.zeros((0, 3), dtype=torch.int64), 'b': full['b']}
    vol = np.zeros(shape, np.float32)
    ref = vt.compute_similarities(vol, feat, full)
    got = vt.compute_similarities(vol, feat, holed)
    assert list(got) == ['a', 'gap', 'b']
    assert torch.equal(got['a'], ref['a']) and torch.equal(got['b'], ref['b'])
    assert got['gap'].dtype == torch.uint8 and got['gap'].shape == ref['a'].shape and int(got['gap'].max()) == 0
    dev = vt.compute_similarities(vol, feat, holed, keep_on_device=True)
    assert all(v.is_cuda for v in dev.values())
    lab = vt.assign_labels(got)
    want = osim.assign_labels([ref['a'], torch.zeros_like(ref['a']), ref['b']])
    assert np.array_equal(lab, want) and set(np.unique(lab)) <= {0, 1, 3}
    assert vt.compute_similarities(vol, feat, {'gap': torch.zeros((0, 3), dtype=torch.int64)}) is None


@pytest.mark.parametrize('f,n', [(64, (6, 7, 8)), (384, (16, 16, 16))])
def test_similarity_single_annotation(gpu, f, n):
    """One query voxel in total (BASELINE configs[0]: '1 similarity query').  The intended map -- the reference's
    .squeeze(1) (predict_ntf.py:65) drops the annotation axis in exactly this case, so the oracle
    computes the mathematically intended map, as the GPU path does."""
    g = gen(f)
    feat = F.normalize(torch.randn(f, *n, generator=g), dim=0)
    feat = F.normalize(feat + 0.6 * feat[:, 2:3, 3:4, 4:5], dim=0).half()
    shape = tuple(2 * s for s in n)
    ann = {'ntf1': torch.tensor([[4, 6, 8]])}
    got = vt.compute_similarities(np.zeros(shape, np.float32), feat, ann)
    ref = osim.similarity_maps(shape, feat.float(), ann)
    assert list(got) == ['ntf1'] and got['ntf1'].shape == ref['ntf1'].shape and int(ref['ntf1'].max()) > 50   # (the maximum voxel itself wraps to 1)
    d = (got['ntf1'].int() - ref['ntf1'].int()).abs()
    d = torch.minimum(d, 256 - d)
    assert int(d.max()) <= 1 and float((d > 0).float().mean()) <= 0.01
    assert np.array_equal(vt.assign_labels(got), osim.assign_labels([got['ntf1']]))


# ---------------------------------------------------------------- fp8 attention (BASELINE configs[3])
def _e4m3(x):
    """Round to OCP e4m3 (torch.float8_e4m3fn) and back: the host-side model of the kernel's operand rounding."""
    return x.to(torch.float8_e4m3fn).to(torch.float64)


@pytest.mark.parametrize('dt', ['fp16', 'bf16'])
@pytest.mark.parametrize('batch,tokens,heads', [(1, 1, 2), (2, 65, 2), (1, 200, 12), (2, 577, 2), (1, 4097, 2)])
def test_attention_fp8(gpu, dt, batch, tokens, heads):
    """vittf_attention_fp8 (e4m3 operands on v_mfma_scale_f32_32x32x64_f8f6f4, per-head power-of-two scales, fp32 softmax
    statistics) against (a) the exact fp64 attention of the same 16-bit inputs -- the stated tolerance of the path: 6e-2
    relative Frobenius, 3-bit mantissas on q, k, v and P -- and (b) a host model that rounds q, k, v to e4m3 with the same
    scales and computes the rest in fp64 -- 2e-2: what is left is the rounding of P and of the output."""
    lib = _lib.load()
    g = gen(tokens + heads)
    d = heads * 64
    qkv = torch.randn(batch * tokens, 3 * d, generator=g)
    qkv[:, :2 * d] *= 1.3
    qkv[:, 2 * d:] = qkv[:, 2 * d:] * 2.0 + 0.5                # v with an offset and its own range
    qkv = _prescale(qkv, heads, dt)
    qd = qkv.to(gpu)
    out = torch.full((batch * tokens + 2, d), 7.0, dtype=TDT[dt], device=gpu)
    ws = torch.empty(lib.vittf_attention_fp8_workspace_bytes(batch, tokens, heads), dtype=torch.uint8, device=gpu)
    _lib.check(lib.vittf_attention_fp8(_lib.ptr(qd), _lib.ptr(out), batch, tokens, heads, _lib.DTYPES[dt], _lib.ptr(ws), ws.numel(),
                                       _lib.stream_ptr()))
    torch.cuda.synchronize()
    got = out.float().cpu().double()
    assert (got[batch * tokens:] == 7.0).all(), 'wrote past the last row'
    got = got[:batch * tokens]
    assert torch.isfinite(got).all()
    exact = _attn_ref(qkv, batch, tokens, heads, 1)
    x = qkv.double().view(batch, tokens, 3, heads, 64)
    xq = torch.empty_like(x)
    for part in range(3):                                       # per (slice, head) power-of-two scale: amax * 2^-e <= 448
        t = x[:, :, part]                                       # (batch, tokens, heads, 64)
        amax = t.abs().amax(dim=(1, 3), keepdim=True).clamp_min(1e-30)
        e = torch.ceil(torch.log2(amax / 448.0))
        e = torch.where(torch.log2(amax / 448.0) == e, e + 1, e)       # frexp convention: mantissa in [0.5, 1)
        xq[:, :, part] = _e4m3(t / 2.0 ** e) * 2.0 ** e
    model = _attn_ref(xq.reshape(batch * tokens, 3 * d), batch, tokens, heads, 1)
    e_exact, e_model = rel_fro(got, exact), rel_fro(got, model)
    print(f'fp8 attention {batch}x{tokens}x{heads} {dt}: rel fro {e_exact:.3e} vs exact, {e_model:.3e} vs the e4m3-operand model')
    assert e_exact <= 6e-2 and e_model <= 2e-2
    assert lib.vittf_attention_fp8(_lib.ptr(qd), _lib.ptr(out), batch, tokens, heads, _lib.DTYPES[dt], _lib.ptr(ws), 16,
                                   _lib.stream_ptr()) == -2      # workspace too small


@pytest.mark.parametrize('dt', ['fp16', 'bf16'])
@pytest.mark.parametrize('batch,tokens', [(1, 1), (3, 65), (2, 200), (1, 4097)])
def test_qkv_fp8_rows_path(gpu, dt, batch, tokens):
    """vittf_gemm_qkv_fp8 + vittf_attention_fp8_rows (ViT-B head count: 12 heads, K = 768): the qkv projection whose q and k
    leave as e4m3 rows with one power-of-two scale per row and 32-wide block (the MX format of the matrix instruction's
    scale operands), v as 16-bit values with its per-(slice, head) maximum collected on the way, then the attention kernel
    with row scales.  Checked: (1) the v third of the qkv buffer = the plain GEMM's bits; (2) the attention output against
    the exact fp64 attention of the exact fp64 projection (the path's stated 6e-2) and against a host model that rounds q and
    k to e4m3 with per-row-block scales and v with its per-(slice, head) scale (2e-2); slices of 65 / 200 tokens put slice
    boundaries inside the GEMM's 256-row tiles."""
    lib = _lib.load()
    heads, k = 12, 768
    d = heads * 64
    n = 3 * d
    rows = batch * tokens
    g = gen(tokens * 7 + batch)
    a = torch.randn(rows, k, generator=g).to(TDT[dt])
    w = (torch.randn(n, k, generator=g) / k ** 0.5)
    w[:2 * d] *= 1.3
    w[2 * d:] *= 2.0
    w = w.to(TDT[dt])
    bias = 0.2 * torch.randn(n, generator=g)
    bias[2 * d:] += 0.5
    ad, wd, bd = a.to(gpu), w.to(gpu), bias.to(gpu)
    # (the workspace starts as 0xff bytes: NaN as e4m3, NaN as an E8M0 scale -- what the padded rows of a tile must not keep)
    ws = torch.full((lib.vittf_attention_fp8_workspace_bytes(batch, tokens, heads),), 0xff, dtype=torch.uint8, device=gpu)
    qkv = torch.full((rows + 2, n), 7.0, dtype=TDT[dt], device=gpu)
    _lib.check(lib.vittf_gemm_qkv_fp8(_lib.ptr(ad), _lib.ptr(wd), _lib.ptr(bd), _lib.ptr(qkv), rows, n, k, tokens, heads,
                                      _lib.DTYPES[dt], _lib.ptr(ws), ws.numel(), _lib.stream_ptr()))
    out = torch.full((rows + 2, d), 7.0, dtype=TDT[dt], device=gpu)
    _lib.check(lib.vittf_attention_fp8_rows(_lib.ptr(qkv), _lib.ptr(out), batch, tokens, heads, _lib.DTYPES[dt], _lib.ptr(ws),
                                            ws.numel(), _lib.stream_ptr()))
    torch.cuda.synchronize()
    # (1) the v third: the same bits as the plain qkv GEMM; nothing behind the last row, nothing in the q / k thirds
    plain = torch.empty(rows, n, dtype=TDT[dt], device=gpu)
    _lib.check(lib.vittf_gemm(_lib.ptr(ad), _lib.ptr(wd), _lib.ptr(bd), _lib.ptr(plain), rows, n, k, _lib.EPI_BIAS_QKV, 0,
                              _lib.DTYPES[dt], _lib.stream_ptr()))
    assert torch.equal(qkv[:rows, 2 * d:], plain[:, 2 * d:])
    assert (qkv[rows:].float() == 7.0).all() and (qkv[:rows, :2 * d].float() == 7.0).all()
    got = out.float().cpu().double()
    assert (got[rows:] == 7.0).all(), 'wrote past the last row'
    got = got[:rows]
    assert torch.isfinite(got).all()
    # (2) exact reference: fp64 projection of the 16-bit operands (+ the q scale), fp64 attention
    proj = (ad.double() @ wd.double().t() + bd.double()).cpu()
    proj[:, :d] *= QSCALE
    # (1b) what the GEMM left in the workspace (layout: attention_fp8.hip::fp8_ws -- amax | q8 | k8 | v8t | qs | ks): every q / k
    # row de-quantised with its block scales is the projection to e4m3 precision (half an ulp of a 3-bit mantissa = 2^-4 of
    # the block maximum at worst), and no block wastes more than one binade of the format's range
    np_ = (tokens + 63) // 64 * 64
    per = batch * heads * np_ * 64
    off_q = (batch * heads * 12 + 255) // 256 * 256
    off_qs = off_q + 3 * per
    sc_bytes = (batch * heads * np_ * 2 + 255) // 256 * 256
    wsc = ws.cpu()
    for part, (o8, osc) in enumerate(((off_q, off_qs), (off_q + per, off_qs + sc_bytes))):
        # a stored row is [d 0-15 | d 32-47 | d 16-31 | d 48-63]: the matrix instruction's MX block b of a row is bytes 16 b ..
        # 16 b + 15 of both lane halves (tools/micro/mfma_f8_scale_probe2.hip), and lane half hh reads bytes 32 hh .. 32 hh + 31
        # rows tokens .. np - 1 of every (slice, head): zero bytes and zero scale bytes (the GEMM used to leave them as found)
        assert (wsc[o8:o8 + per].view(batch, heads, np_, 64)[:, :, tokens:] == 0).all(), 'padded q / k rows'
        assert (wsc[osc:osc + batch * heads * np_ * 2].view(batch, heads, np_, 2)[:, :, tokens:] == 0).all(), 'padded scale rows'
        q8 = wsc[o8:o8 + per].view(torch.float8_e4m3fn).to(torch.float64).view(batch, heads, np_, 2, 2, 16)[:, :, :tokens]
        q8 = q8.permute(0, 1, 2, 4, 3, 5).reshape(batch, heads, tokens, 2, 32)            # (lane half, block, 16) -> (block, 32)
        sc = wsc[osc:osc + batch * heads * np_ * 2].view(batch, heads, np_, 2)[:, :, :tokens].to(torch.float64)
        deq = q8 * (2.0 ** (sc - 127.0))[..., None]                                   # (batch, heads, tokens, 2, 32)
        want = proj.view(batch, tokens, 3, heads, 2, 32)[:, :, part].permute(0, 2, 1, 3, 4)
        bmax = want.abs().amax(dim=-1, keepdim=True)
        assert ((deq - want).abs() <= bmax * 2.0 ** -4 + 1e-6).all(), f'part {part}: de-quantised rows differ from the projection'
        assert (q8.abs().amax(dim=-1) >= 224.0 * (bmax[..., 0] > 1e-6)).all(), f'part {part}: a block scale wastes range'
    exact = _attn_ref(proj, batch, tokens, heads, 1)
    # model: q, k rounded to e4m3 per (row, 32-wide block) with 2^e scales; v from its 16-bit values with the (slice, head) scale
    x = proj.view(batch, tokens, 3, heads, 2, 32).clone()
    for part in range(2):
        t = x[:, :, part]                                          # (batch, tokens, heads, 2, 32)
        amax = t.abs().amax(dim=-1, keepdim=True).clamp_min(1e-30)
        e = torch.ceil(torch.log2(amax / 448.0))
        e = torch.where(torch.log2(amax / 448.0) == e, e + 1, e).clamp_min(-20)
        x[:, :, part] = _e4m3(t / 2.0 ** e) * 2.0 ** e
    v16 = plain[:, 2 * d:].float().cpu().double().view(batch, tokens, heads, 64)
    amax = v16.abs().amax(dim=(1, 3), keepdim=True).clamp_min(1e-30)
    e = torch.ceil(torch.log2(amax / 448.0))
    e = torch.where(torch.log2(amax / 448.0) == e, e + 1, e)
    x[:, :, 2] = (_e4m3(v16 / 2.0 ** e) * 2.0 ** e).view(batch, tokens, heads, 2, 32)
    model = _attn_ref(x.reshape(rows, 3 * d), batch, tokens, heads, 1)
    e_exact, e_model = rel_fro(got, exact), rel_fro(got, model)
    print(f'fp8 rows path {batch}x{tokens}x{heads} {dt}: rel fro {e_exact:.3e} vs exact, {e_model:.3e} vs the MX-operand model')
    assert e_exact <= 6e-2 and e_model <= 2e-2
    assert lib.vittf_gemm_qkv_fp8(_lib.ptr(ad), _lib.ptr(wd), _lib.ptr(bd), _lib.ptr(qkv), rows, n, k, tokens, heads,
                                  _lib.DTYPES[dt], _lib.ptr(ws), 16, _lib.stream_ptr()) == -2      # workspace too small
    assert lib.vittf_gemm_qkv_fp8(_lib.ptr(ad), _lib.ptr(wd), _lib.ptr(bd), _lib.ptr(qkv), rows, n, k, tokens, heads + 1,
                                  _lib.DTYPES[dt], _lib.ptr(ws), ws.numel(), _lib.stream_ptr()) == -1     # n != 3 * heads * 64


def test_non_finite_volume_is_refused(gpu):
    vol = torch.rand((8, 8, 8), generator=gen(1))
    vol[3, 4, 5] = float('nan')
    with pytest.raises(ValueError):
        vt.DeviceVolume(vol, gpu)
    vol[3, 4, 5] = float('inf')
    with pytest.raises(ValueError):
        vt.DeviceVolume(vol.half(), gpu)
