"""CPU, world_size 2 over gloo: the slice-sharding + all-gather + assemble logic of vit_tf_amd.extract, with the
device kernels replaced by an oracle-backed stand-in (tests only -- the product has no such path).
Checks that 2 ranks reproduce, bit for bit, what the single-process oracle computes."""
import os
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))


class OracleOps:
    """CPU stand-ins with the exact semantics of the libvittf kernels the sharding logic calls."""

    def __init__(self, oracle_model):
        self.m = oracle_model

    def volume(self, vol, model):
        class V:
            pass
        v = V(); v.data = torch.as_tensor(vol).float().squeeze(); v.shape = tuple(v.data.shape)
        return v

    def zeros(self, shape, model):
        return torch.zeros(shape, dtype=torch.float16)

    def k_slices(self, model, dvol, axis, im_sizes, s0, s1, engine_batch, part):
        from oracle import feature_volume as ofv
        imgs = ofv.normalized_slices(dvol.data, axis)[s0:s1]
        rows, cols = ofv.axis_image_size(im_sizes, axis)
        with torch.no_grad():      # one slice per forward: CPU GEMM bits depend on the batch shape
            x = torch.nn.functional.interpolate(imgs, size=(rows, cols), mode='nearest')
            return torch.cat([ofv.k_tokens(self.m, x[i:i + 1]).half()[:, 1:] for i in range(x.shape[0])]).contiguous()

    def pool(self, model, kbuf, k_s0, n_total, n_out, win0, nwin, f0, f1, d, dst, strides):
        from vit_tf_amd.extract import window_bounds
        sd, sw, sr, sc = strides
        flat = dst.view(-1)
        k = kbuf.view(kbuf.shape[0], f0, f1, d).float()
        for i in range(nwin):
            lo, hi = window_bounds(win0 + i, n_total, n_out)
            acc = torch.zeros(f0, f1, d)
            for s in range(lo, hi):                     # running sum kept in fp16, like the CPU pooling kernel
                acc = (acc + k[s - k_s0]).half().float()
            mean = (acc / float(hi - lo)).half()
            idx = (torch.arange(d).view(1, 1, d) * sd + i * sw + torch.arange(f0).view(f0, 1, 1) * sr
                   + torch.arange(f1).view(1, f1, 1) * sc)
            flat[idx.reshape(-1)] = mean.reshape(-1)

    def assemble_sum(self, model, gz, gy, gx, world, chunks, d, feat_out):
        n0, n1, n2 = feat_out
        z = torch.cat(list(gz), dim=3)[:, :, :, :n2]
        y = torch.cat(list(gy), dim=2)[:, :, :n1]
        x = torch.cat(list(gx), dim=1)[:, :n0]
        return ((0.0 + z) + y) + x


class FakeModel:
    embed_dim, patch_size, device = 128, 8, torch.device('cpu')


def _worker(rank, world, port, shape, fos, seed, q):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    torch.set_num_threads(1)
    try:
        import vit_tf_amd as vt
        from helpers import tiny_model
        if world == 1:       # a ONE-rank group still runs every call of the exchange (VITTF_DIST_FORCE=1 sets this at import)
            vt.extract.DIST_FORCE = True
        oracle, _ = tiny_model(seed)
        vol = (torch.rand(shape, generator=torch.Generator().manual_seed(1)) * 2 - 1).half().float()
        ops = OracleOps(oracle)
        full = vt.feature_volume(vol, FakeModel(), fos, 'all', ops=ops)
        single = vt.feature_volume(vol, FakeModel(), fos, 'y', ops=ops)
        if world == 1:
            assert vt.extract.EXCHANGES == {'gloo': 4}, vt.extract.EXCHANGES       # z, y, x of 'all' + the single axis
        if rank == 0:
            q.put((full.numpy(), single.numpy()))       # by value: the producer may exit before the consumer reads
    finally:
        dist.barrier()
        dist.destroy_process_group()


def _free_port():
    import socket
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        return s.getsockname()[1]


# world 4 with 3 / 4 pooling windows per axis: uneven chunks, and (fos 3) a rank that owns no window at all -- the shape of
# an 8-GPU run whose window count is not a multiple of the rank count.  World 8: (512, 16, 16) at fos 2 has the benchmark's
# own ratio along x -- 512 slices -> 64 windows of 8 slices -> 8 windows per rank -- while its y and z axes have 2 windows
# for 8 ranks (six ranks contribute an empty slab); (80, 16, 16): 10 windows over 8 ranks, chunk 2, three ranks without one.
@pytest.mark.parametrize('shape,fos,world', [((24, 16, 32), 3, 2), ((10, 10, 10), 4, 2), ((24, 16, 32), 3, 4), ((10, 10, 10), 4, 4),
                                             ((512, 16, 16), 2, 8), ((80, 16, 16), 2, 8), ((10, 10, 10), 4, 1)])
def test_two_rank_sharding_matches_single_process_oracle(shape, fos, world):
    from oracle import feature_volume as ofv
    from helpers import tiny_model
    seed = 3
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, shape, fos, seed, q)) for r in range(world)]
    for p in procs:
        p.start()
    full, single = (torch.from_numpy(a) for a in q.get(timeout=300))
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    oracle, _ = tiny_model(seed)
    vol = (torch.rand(shape, generator=torch.Generator().manual_seed(1)) * 2 - 1).half().float()
    threads = torch.get_num_threads()
    torch.set_num_threads(1)
    try:
        ref_full = ofv.feature_volume(vol, oracle, 8, fos, 'all', batch_size=1)
        ref_single = ofv.feature_volume(vol, oracle, 8, fos, 'y', batch_size=1)
    finally:
        torch.set_num_threads(threads)
    assert full.dtype == torch.float16 and torch.equal(full, ref_full)
    assert torch.equal(single, ref_single)
