"""Feature-volume extraction: volume -> (F, W', H', D') fp16 K-feature volume, on one or several GPUs.

Host-side mirror of infer.py:130-210 (compute_qkv) and infer.py:314-333 (the __main__ driver): the
sizing rule, the per-axis slice views, the slice loop, slice-axis pooling and the z -> y -> x fp16 sum.
All arithmetic happens in libvittf kernels; this module only sizes buffers, walks slice ranges and (for
more than one rank) exchanges pooled slabs with ONE all-gather per axis over RCCL.

Sharding: slices of an axis are independent, so rank r of G owns the contiguous
block of pooling windows [r*c, (r+1)*c), c = ceil(n_out / G), runs exactly the slices those windows
touch, pools them locally into a slab, and the slabs are all-gathered; every rank then forms
fp16(fp16(z + y) + x) for the whole volume.
"""
import ctypes as C

import torch

from . import _lib

# axis -> (sliced volume dim, (volume dim of image rows, volume dim of image cols))      infer.py:138-147
AXIS_DIMS = {'z': (2, (0, 1)), 'y': (1, (0, 2)), 'x': (0, (1, 2))}
# Slices per vittf_vit_k_features call.  Slices are independent and results do not depend on the batching (tested: 31 / 32 /
# 256 / 512 give the same bits); larger batches amortise every launch's partial last round of workgroups.  The persistent
# kernels of the ViT-S path hand out 128- / 256-row tiles to 256 CUs: 256 slices x 4097 tokens are 16.004 rounds of the qkv
# projection's tiles and 32.01 of the block tail's -- a seventeenth / thirty-third round for one tile in 256 --, 512 slices
# halve that waste: 2318 -> 2340 slices/s on one box (block tail 3.07 -> 2.96 ms per 256 slices, qkv 0.989 -> 0.976; round 5).
# 512 slices of N = 4097 tokens at D = 384 = 11 GB of workspace (of 288 GB).  D = 768 stays at 256: its widest buffer (rows x
# 4 D 16-bit values) must stay below 2^32 elements.
DEFAULT_ENGINE_BATCH = 512
DEFAULT_ENGINE_BATCH_WIDE = 256       # embed_dim > 384
MAX_ENGINE_BATCH_ROWS = 1024 * 4097   # what an AtLeast request may raise a call to (32-bit offsets into the widest buffers)


class AtLeast(int):
    """An engine-batch request that is a LOWER bound: what the reference's `--batch-size` / `compute_qkv(batch_size=...)`
    become here.  The reference's mini-batch bounds its GPU memory; this engine sizes its own calls (a 4-slice launch is
    ~128 row tiles for 256 CUs and runs several times slower for the same bits), so the flag may raise the engine's
    batch but never lowers it.  A plain int (bench.py --engine-batch, the tests) is taken literally."""


def engine_batch_for(tokens, embed_dim, requested=None):
    """Slices per engine call.  The default keeps the ROW count of a call at what 512 slices of N = 4097 are for D <= 384
    (2.1 M rows: 11 GB of workspace) and 256 slices for wider models (1.05 M rows: 11 GB at D = 768; the widest buffer, rows x
    4 D 16-bit values, stays below 2^32 elements), whatever the token count: base * 4097 / tokens, clamped to 1 .. base -- the
    fos-128 preset (N = 16385) then runs 128 slices per call.  `requested`: a plain int (bench.py --engine-batch, tests) is
    taken literally, an `AtLeast` (infer.py --batch-size, compute_qkv's batch_size) only raises the default;
    VITTF_ENGINE_BATCH overrides both; results never depend on any of them."""
    env = __import__('os').environ.get('VITTF_ENGINE_BATCH')
    if env:
        return max(1, int(env))
    base = DEFAULT_ENGINE_BATCH if int(embed_dim) <= 384 else DEFAULT_ENGINE_BATCH_WIDE
    default = max(1, min(base, base * 4097 // int(tokens)))
    if isinstance(requested, AtLeast):
        # (capped: the widest buffers of a call -- rows x 4 D 16-bit values, the fp8 operand rows -- are addressed with 32-bit offsets)
        return min(max(int(requested), default), max(default, MAX_ENGINE_BATCH_ROWS // int(tokens)))
    if requested:
        return max(1, int(requested))
    return default


PARTS = {'q': 0, 'k': 1, 'v': 2}
# VITTF_DIST_FORCE=1: run the slab exchange even when the process group has ONE rank (the collective then moves nothing, but
# every call of the multi-rank path -- the in-place all_gather_into_tensor, its deferred wait -- executes on the backend)
DIST_FORCE = __import__('os').environ.get('VITTF_DIST_FORCE', '0') == '1'


def sizing(vol_shape, feature_output_size, patch_size):
    """infer.py:317-319: (im_sz, feat_out_sz)."""
    ref_fact = sorted(vol_shape[-3:])[1] / feature_output_size
    im_sz = tuple(int(patch_size * (d // ref_fact)) for d in vol_shape[-3:])
    return im_sz, tuple(d // patch_size for d in im_sz)


def window_bounds(i, n_in, n_out):
    """Adaptive-pool window i: [floor(i*in/out), ceil((i+1)*in/out))."""
    return (i * n_in) // n_out, -((-(i + 1) * n_in) // n_out)


def shard_windows(n_out, rank, world):
    """Windows owned by `rank`: (first, count, chunk) with chunk = ceil(n_out / world)."""
    chunk = -(-n_out // world)
    first = min(rank * chunk, n_out)
    return first, max(0, min(chunk, n_out - first)), chunk


class DeviceVolume:
    """fp32 copy of the volume in HBM + its global min/max (norm_minmax, infer.py:32-34, 137)."""

    def __init__(self, vol, device):
        lib = _lib.require_device()
        vol = torch.as_tensor(vol).squeeze()
        if vol.ndim != 3:
            raise ValueError(f'volume must be 3-D, got shape {tuple(vol.shape)}')
        self.shape = tuple(vol.shape)
        if vol.dtype == torch.float16 and not vol.is_cuda:
            # volumes are stored as fp16 (create_synthetic_volumes.py:44-46): upload the 2-byte form and widen it in HBM
            # (vol.float(), infer.py:137) -- half the PCIe bytes of a host-side conversion
            raw = vol.contiguous().to(device=device)
            self.data = torch.empty(self.shape, dtype=torch.float32, device=device)
            _lib.check(lib.vittf_widen_f16(_lib.ptr(raw), raw.numel(), _lib.ptr(self.data), _lib.stream_ptr()),
                       'vittf_widen_f16')
            del raw
        else:
            self.data = vol.to(device=device, dtype=torch.float32).contiguous()
        if not bool(torch.isfinite(self.data).all()):
            # the attention kernels are built without NaN handling (scores are never NaN for finite input): refuse loudly
            # instead of returning a feature volume of garbage
            raise ValueError('volume contains NaN or Inf voxels')
        self.minmax = torch.empty(2, dtype=torch.float32, device=device)
        ws = torch.empty(lib.vittf_minmax_workspace_bytes(), dtype=torch.uint8, device=device)
        _lib.check(lib.vittf_volume_minmax(_lib.ptr(self.data), self.data.numel(), _lib.ptr(self.minmax), _lib.ptr(ws),
                                           ws.numel(), _lib.stream_ptr()), 'vittf_volume_minmax')
        self._ws = ws

    def view(self, axis, im_sizes):
        sl, (a, b) = AXIS_DIMS[axis]
        st = self.data.stride()
        return _lib.SliceView(self.data.data_ptr(), st[sl], st[a], st[b], self.shape[a], self.shape[b],
                              im_sizes[a], im_sizes[b], self.minmax.data_ptr())


def _axis_geometry(shape, im_sizes, axis, patch):
    sl, (a, b) = AXIS_DIMS[axis]
    return sl, a, b, shape[sl], im_sizes[a] // patch, im_sizes[b] // patch


def k_slices(model, dvol, axis, im_sizes, s0, s1, engine_batch=None, part=1, out=None):
    """Token-major fp16 features of slices [s0, s1) of one axis: tensor [s1-s0, f0*f1, D] on the device."""
    _, _, _, _, f0, f1 = _axis_geometry(dvol.shape, im_sizes, axis, model.patch_size)
    engine_batch = engine_batch_for(f0 * f1 + 1, model.embed_dim, engine_batch)
    n = s1 - s0
    per = f0 * f1 * model.embed_dim
    if out is None:
        out = torch.empty((n, f0 * f1, model.embed_dim), dtype=torch.float16, device=model.device)
    view = dvol.view(axis, im_sizes)
    flat = out.view(-1)
    for b0 in range(0, n, engine_batch):
        model.k_features(view, s0 + b0, min(engine_batch, n - b0), flat[b0 * per:], part)
    return out


class HipOps:
    """The device operations the sharding logic below is written against: all of them libvittf kernels.
    (tests/ substitute a CPU stand-in built on the oracle to exercise the multi-rank logic under gloo;
    the product never does.)"""

    def volume(self, vol, model):
        return DeviceVolume(vol, model.device)

    def k_slices(self, model, dvol, axis, im_sizes, s0, s1, engine_batch, part):
        return k_slices(model, dvol, axis, im_sizes, s0, s1, engine_batch, part)

    def zeros(self, shape, model):
        return torch.zeros(shape, dtype=torch.float16, device=model.device)

    def pool(self, model, kbuf, k_s0, n_slices_total, n_out, win0, nwin, f0, f1, d, dst, strides):
        sd, sw, sr, sc = strides
        _lib.check(model.lib.vittf_pool_slices(_lib.ptr(kbuf), k_s0, kbuf.shape[0], n_slices_total, n_out, win0, nwin,
                                               f0, f1, d, _lib.ptr(dst), sd, sw, sr, sc, _lib.stream_ptr()),
                   'vittf_pool_slices')

    def assemble_sum(self, model, gz, gy, gx, world, chunks, d, feat_out):
        out = torch.empty((d, *feat_out), dtype=torch.float16, device=model.device)
        carr = (C.c_int32 * 3)(*chunks)
        _lib.check(model.lib.vittf_assemble_sum(_lib.ptr(gz), _lib.ptr(gy), _lib.ptr(gx), world, carr, d, feat_out[0],
                                                feat_out[1], feat_out[2], _lib.ptr(out), _lib.stream_ptr()),
                   'vittf_assemble_sum')
        return out


_HIP_OPS = HipOps()


def _slab_shape_strides(axis, d, n, chunk):
    """Slab of one rank for `axis`: the pooled volume with the axis' dim cut to `chunk` windows.
    Returns (shape, (stride_d, stride_win, stride_row, stride_col)) in elements."""
    sl, (a, b) = AXIS_DIMS[axis]
    dims = list(n)
    dims[sl] = chunk
    st = [dims[1] * dims[2], dims[2], 1]
    return (d, *dims), (dims[0] * dims[1] * dims[2], st[sl], st[a], st[b])


def axis_features(model, dvol, axis, im_sizes, n_out, engine_batch=None, part=1, group=None,
                  ops=_HIP_OPS, pending=None):
    """Pooled (n_out windows along the slice dim) features of one axis, gathered over the process group.

    Returns (gathered [world, D, *slab_dims] fp16 device tensor, chunk).  With a `pending` list the exchange is only
    enqueued (RCCL runs it on its own stream while the next axis is computed) and the caller finishes it with
    `finish_exchanges(pending)` before it reads `gathered`."""
    world = torch.distributed.get_world_size(group) if _dist_on(group) else 1
    rank = torch.distributed.get_rank(group) if _dist_on(group) else 0
    sl, a, b, n_slices, f0, f1 = _axis_geometry(dvol.shape, im_sizes, axis, model.patch_size)
    d = model.embed_dim
    n = [0, 0, 0]
    n[sl], n[a], n[b] = n_out, f0, f1
    win0, nwin, chunk = shard_windows(n_out, rank, world)
    shape, strides = _slab_shape_strides(axis, d, n, chunk)
    gathered = ops.zeros((world, *shape), model)
    slab = gathered[rank]
    if nwin > 0:
        s0 = window_bounds(win0, n_slices, n_out)[0]
        s1 = window_bounds(win0 + nwin - 1, n_slices, n_out)[1]
        kbuf = ops.k_slices(model, dvol, axis, im_sizes, s0, s1, engine_batch, part)
        ops.pool(model, kbuf, s0, n_slices, n_out, win0, nwin, f0, f1, d, slab, strides)
        del kbuf
    if _dist_on(group):
        handle = _all_gather_slabs(gathered, slab, group, defer=pending is not None)   # the one exchange step per axis
        if handle is not None:
            pending.append(handle)
    return gathered, chunk


def finish_exchanges(pending):
    """Make the current stream wait for the all-gathers enqueued with `pending` (no host synchronisation)."""
    for work, _keep_alive in pending:
        work.wait()
    pending.clear()


EXCHANGES = {}       # backend -> slab exchanges enqueued by this process (infer.py prints it; the one-rank RCCL test reads it)


def _all_gather_slabs(gathered, slab, group, defer=False):
    """All-gather the ranks' pooled slabs into `gathered` ([world, ...], this rank's slab already in place).
    RCCL (backend 'nccl') takes the single-tensor form -- asynchronously when `defer` is set: the (work, send buffer)
    pair is returned and the collective overlaps whatever the caller enqueues next; gloo (CPU tests, or a
    2-ranks-on-1-GPU rehearsal) gets the list form, staged through the host when the backend cannot take device tensors."""
    world = gathered.shape[0]
    flat = gathered.view(world, -1)
    backend = torch.distributed.get_backend(group)
    EXCHANGES[backend] = EXCHANGES.get(backend, 0) + 1
    if backend == 'nccl':
        # in place: this rank's slab already sits at flat[rank], which is exactly where an all-gather writes the rank's own
        # contribution (send buffer = receive buffer + rank * count, RCCL's in-place form): no staging copy
        mine = flat[torch.distributed.get_rank(group)]
        if slab.data_ptr() != mine.data_ptr():
            raise RuntimeError('in-place all-gather: this rank\'s slab is not its slot of the receive buffer')
        if defer:
            return torch.distributed.all_gather_into_tensor(flat.view(-1), mine, group=group, async_op=True), mine
        torch.distributed.all_gather_into_tensor(flat.view(-1), mine, group=group)
        return None
    mine = slab.reshape(-1).clone()       # gloo: the output list must not alias the input
    if flat.is_cuda:
        host = [torch.empty(flat.shape[1], dtype=flat.dtype) for _ in range(world)]
        torch.distributed.all_gather(host, mine.cpu(), group=group)
        for r in range(world):
            flat[r].copy_(host[r])
    else:
        torch.distributed.all_gather(list(flat.unbind(0)), mine, group=group)
    return None


def _dist_on(group):
    return torch.distributed.is_available() and torch.distributed.is_initialized() and \
        (torch.distributed.get_world_size(group) > 1 or DIST_FORCE)


def assemble_axis(gathered, axis, n_total):
    """Single-axis mode: concatenate the ranks' slabs along the slice dim -> (D, ...) (plumbing only)."""
    sl, _ = AXIS_DIMS[axis]
    full = torch.cat(list(gathered), dim=1 + sl)
    return full.narrow(1 + sl, 0, n_total).contiguous()


def feature_volume(vol, model, feature_output_size=64, slice_along='all', engine_batch=None,
                   part=1, group=None, dvol=None, ops=_HIP_OPS):
    """infer.py:314-333 on the GPU(s).  Returns the fp16 feature tensor on the device:
    'all' -> (D, W', H', D') = fp16(fp16(z + y) + x) of the pooled axes; 'x'|'y'|'z' -> un-pooled single axis."""
    if dvol is None:
        dvol = ops.volume(vol, model)
    im_sz, feat_out = sizing(dvol.shape, feature_output_size, model.patch_size)
    if min(im_sz) <= 0:
        raise ValueError(f'feature_output_size {feature_output_size} gives an empty image size {im_sz}')
    if slice_along in AXIS_DIMS:
        sl = AXIS_DIMS[slice_along][0]
        g, _ = axis_features(model, dvol, slice_along, im_sz, dvol.shape[sl], engine_batch, part, group, ops)
        return assemble_axis(g, slice_along, dvol.shape[sl])
    if slice_along != 'all':
        raise Exception(f'Invalid argument for --slice-along: {slice_along}. Must be x,y,z or all')
    gathered, chunks, pending = {}, [0, 0, 0], []
    for ax in ('z', 'y', 'x'):
        sl = AXIS_DIMS[ax][0]
        gathered[ax], chunks[sl] = axis_features(model, dvol, ax, im_sz, feat_out[sl], engine_batch, part, group, ops,
                                                 pending=pending)
    finish_exchanges(pending)        # the z and y exchanges have run under the compute of the following axes
    world = gathered['z'].shape[0]
    out = ops.assemble_sum(model, gathered['z'], gathered['y'], gathered['x'], world, chunks, model.embed_dim, feat_out)
    return out.squeeze()        # the reference's running sum drops singleton dims (infer.py:332 v.squeeze())


def pooled_axis(vol, model, axis, im_sizes, out_size, engine_batch=None, part=1, group=None, dvol=None,
                ops=_HIP_OPS):
    """compute_qkv(..., pool_fn=AdaptiveAvgPool3d(out_size)) for one axis: (D, *out_size) fp16 on the device."""
    if dvol is None:
        dvol = ops.volume(vol, model)
    sl, (a, b) = AXIS_DIMS[axis]
    p = model.patch_size
    if out_size[a] != im_sizes[a] // p or out_size[b] != im_sizes[b] // p:
        raise NotImplementedError('in-plane pooling: the HIP path pools the slice axis only (the token grid already '
                                  'equals feat_out_sz for every sizing infer.py produces, infer.py:317-319)')
    g, _ = axis_features(model, dvol, axis, im_sizes, out_size[sl], engine_batch, part, group, ops)
    return assemble_axis(g, axis, out_size[sl])
