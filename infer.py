"""Drop-in for the reference's ``infer.py``: slice-wise DINO ViT K-feature volumes, MI355X-native.

Same module surface and CLI as /root/reference/infer.py (helpers :10-40, sample_features3d :48,
compute_qkv :130, load_data :212, load_model :239, handle_output_path :266, CLI flags :296-306,
output naming :279, file layout :337-340) -- the ViT forward, pooling and axis sum run in the HIP
kernels of libvittf instead of stock PyTorch ops.  There is no CPU path: ``--cpu`` is accepted for flag
compatibility and refused with exit code 1.

Differences a caller can observe:
  * the model comes from a LOCAL DINO state dict (``--weights``, ``$VITTF_WEIGHTS`` or the torch-hub
    checkpoint cache) or from seeded synthetic weights (``--synthetic-weights SEED``); nothing is
    fetched from the network (the reference calls torch.hub.load, :42-43)
  * ``--batch-size`` no longer changes memory behaviour: slices are independent and the engine picks its
    own batch (results do not depend on it); ``--engine-dtype`` selects fp16 (default, the reference's autocast type :309) or bf16 MFMA operands
  * launched under torchrun, the slices of each axis are sharded over the ranks and reassembled with one
    RCCL all-gather per axis; rank 0 writes the file
"""
import os
import sys
import time
from collections import defaultdict
from pathlib import Path

import numpy as np
import torch

import vit_tf_amd as vt

in_mean = [0.485, 0.456, 0.406]
in_std = [0.229, 0.224, 0.225]


# ---------------------------------------------------------------------------- tensor helpers (:10-37)
def make_nd(t, n):
    """Prepend singleton dims until `t` is n-dimensional."""
    if n < t.ndim:
        raise Exception(f'make_nd cannot reduce cardinality. Your Tensor.ndim={t.ndim} > n={n}.')
    return t[(None,) * (n - t.ndim)] if n > t.ndim else t


def make_3d(t):
    return make_nd(t, 3)


def make_4d(t):
    return make_nd(t, 4)


def make_5d(t):
    return make_nd(t, 5)


def norm_minmax(t):
    lo, hi = t.min(), t.max()
    return (t - lo) / (hi - lo)


def norm_mean_std(t, mu=0, std=1):
    tf = t.float()
    return (tf - tf.mean()) * std / tf.std() + mu


# ---------------------------------------------------------------------------- model (:42-46, :239-264)
_MODEL_OPTS = {'weights': None, 'synthetic_seed': None, 'dtype': 'fp16', 'attention': '16bit'}


def get_dino_model(name):
    """HIP engine for ``dino_<name>``.  Weight source, in order: --weights / $VITTF_WEIGHTS, the local
    torch-hub checkpoint cache, --synthetic-weights / $VITTF_SYNTHETIC_WEIGHTS.  No network access."""
    path = _MODEL_OPTS['weights'] or vt.find_local_checkpoint(name)
    seed = _MODEL_OPTS['synthetic_seed']
    if seed is None and os.environ.get('VITTF_SYNTHETIC_WEIGHTS') is not None:
        seed = int(os.environ['VITTF_SYNTHETIC_WEIGHTS'])
    if path is not None:
        sd = vt.load_state_dict_file(path)
    elif seed is not None:
        print(f'Using seeded synthetic {name} weights (seed {seed}); features are NOT DINO features.')
        sd = vt.synthetic_state_dict(name, seed)
    else:
        print(f'No local checkpoint for dino_{name}: pass --weights PATH (DINO state dict) or '
              f'--synthetic-weights SEED. This build never downloads weights.')
        sys.exit(1)
    return vt.HipViT(sd, name, dtype=_MODEL_OPTS['dtype'], attention=_MODEL_OPTS['attention'])


def get_dinov2_model(name):
    print('DINOv2 (patch 14) models are not supported by the HIP engine.')
    sys.exit(1)


def load_model(args):
    """(:239-264) -> (model name, constructor, patch size); exits with 1 on contradictory flags."""
    if args.dino_model and args.dino2_model:
        print('Both --dino-model and --dino2-model were set. Please only set one of them.')
        sys.exit(1)
    if args.dino2_model:
        args.dino_model = args.model = args.dino2_model
        return args.dino2_model, get_dinov2_model, 14
    if not args.dino_model:
        print('No DINO/DINOv2 model specified, using default: vits8')
        args.dino_model = 'vits8'
    args.model = args.dino_model
    return args.dino_model, get_dino_model, 8 if args.dino_model[-1] == '8' else 16


# ---------------------------------------------------------------------------- sampling helpers (:48-126)
def sample_features3d(feat_vol, rel_coords, mode='nearest'):
    """(:48-72) feat_vol ([M,] F, W, H, D), rel_coords ([M,] C, A, 3) -> ([M,] C, A, F); on the GPU."""
    return vt.sample_features3d(feat_vol, rel_coords, mode)


def resample_topk(feat_vol, sims, K=8, similarity_exponent=2.0, feature_sampling_mode='nearest'):
    """(:75-106) top-K voxels per (class, annotation) map -> re-sampled queries -> mean clamp(sim, 0, 1) ** exponent."""
    return vt.similarity.resample_topk(feat_vol, sims, K, similarity_exponent, feature_sampling_mode)


def take_most_dissimilar(features, num_prototypes=35, measure='cosine'):
    """(:108-126) the num_prototypes rows with the largest mean cosine / euclidean distance to all rows."""
    return vt.similarity.take_most_dissimilar(features, num_prototypes, measure)


def _noop(x, **kwargs):
    return x


# ---------------------------------------------------------------------------- extraction (:130-210)
def compute_qkv(vol, model, patch_size, im_sizes, pool_fn=_noop, batch_size=1, slice_along='z',
                return_keys=['q', 'k', 'v'], dev=None, typ=None, group=None):
    """(:130-210) features of the last block's qkv projection for every slice of one axis.

    Returns {key: CPU tensor} with, like the reference, (F, W', H', S)-style un-pooled layout for
    ``pool_fn=_noop`` and the pooled (F, *output_size) volume when ``pool_fn`` is an
    ``AdaptiveAvgPool3d``; any other callable is applied to the un-pooled device tensor.
    `model` is a HipViT; `dev` / `typ` are accepted for signature compatibility (the engine's device
    and MFMA dtype are properties of the model)."""
    if isinstance(return_keys, str):
        return_keys = [return_keys]
    if patch_size != model.patch_size:
        raise ValueError(f'patch_size {patch_size} != model patch size {model.patch_size}')
    dvol = vt.DeviceVolume(vol, model.device)
    eb = vt.extract.AtLeast(batch_size) if int(batch_size) > 1 else None      # a lower bound: the engine sizes its own calls
    out = {}
    for key in return_keys:
        part = vt.extract.PARTS[key]
        if isinstance(pool_fn, torch.nn.AdaptiveAvgPool3d):
            size = pool_fn.output_size
            size = (size,) * 3 if isinstance(size, int) else tuple(size)
            res = vt.pooled_axis(None, model, slice_along, tuple(im_sizes), size, eb, part, group, dvol)
        else:
            sl = vt.AXIS_DIMS[slice_along][0]
            g, _ = vt.extract.axis_features(model, dvol, slice_along, tuple(im_sizes), dvol.shape[sl], eb, part, group)
            res = pool_fn(vt.extract.assemble_axis(g, slice_along, dvol.shape[sl]))
        out[key] = res.cpu()
    return out


# ---------------------------------------------------------------------------- I/O (:212-237, :266-288)
def load_data(data_path, keep_dtype=False):
    """(:212-237).  keep_dtype=True (used by main()): a .npy volume stays in its stored type instead of fp32, so that an
    fp16 file is uploaded as 2-byte values and widened on the GPU (vittf_widen_f16) -- same fp32 values."""
    data_path = Path(data_path)
    if not data_path.exists():
        print(f'Invalid argument for --data-path (File does not exist): {data_path}')
        sys.exit(1)
    print(f'Attempting to load {data_path}.')
    if data_path.suffix in ('.pt', '.pth'):
        data = torch.load(data_path, weights_only=False)
        vol = data['vol'] if isinstance(data, dict) else data
    elif data_path.suffix == '.npy':
        data = np.load(data_path, allow_pickle=True)
        arr = data[()]['vol'] if data.dtype == object else data
        arr = np.asarray(arr)
        vol = torch.from_numpy(arr if keep_dtype and arr.dtype == np.float16 else arr.astype(np.float32))
    else:
        print(f'Unsupported file extension: {data_path.suffix}')
        sys.exit(1)
    print(f'Loaded volume: {vol.shape} of type {vol.dtype}.')
    assert vol.ndim == 3
    return vol


def handle_output_path(args):
    data_path = Path(args.data_path)
    if not args.cache_path:
        stem = f'{data_path.stem}_{args.model.replace("/", "_")}_{args.slice_along}_features{args.feature_output_size}'
        args.cache_path = data_path.parent / f'{stem}{data_path.suffix}'
    cache_path = Path(args.cache_path)
    if cache_path.exists() and not args.overwrite:
        print(f'Cache file already exists: {cache_path}. Use --overwrite to overwrite.')
        sys.exit(1)
    if not os.access(os.path.dirname(str(cache_path)) or os.getcwd(), os.W_OK):
        print(f'Invalid argument for --cache-path (Cannot write to location): {args.cache_path}')
        sys.exit(1)
    return cache_path


def save_features(qkv, cache_path):
    """(:337-340) {'k': fp16 (F, W', H', D')} as .pt (torch.save) or .npy (pickled dict)."""
    cache_path = Path(cache_path)
    if cache_path.suffix in ('.pt', '.pth'):
        torch.save(qkv, cache_path)
    elif cache_path.suffix == '.npy':
        np.save(cache_path, {k: v.numpy() for k, v in qkv.items()})


def _init_distributed():
    """One process per GPU under torchrun (RANK/LOCAL_RANK/WORLD_SIZE); single process otherwise.  VITTF_DIST_FORCE=1 opens
    the process group even for ONE rank (RANK / WORLD_SIZE / MASTER_* from the environment as usual), so that the slab
    exchange -- in-place all_gather_into_tensor on RCCL, async, waited for in finish_exchanges -- runs on a one-GPU box."""
    world = int(os.environ.get('WORLD_SIZE', '1'))
    if world <= 1 and not vt.extract.DIST_FORCE:
        return 0, 1
    local = int(os.environ.get('LOCAL_RANK', '0')) % max(1, torch.cuda.device_count())
    torch.cuda.set_device(local)
    if not torch.distributed.is_initialized():
        backend = os.environ.get('VITTF_DIST_BACKEND', 'nccl')      # 'nccl' is RCCL on ROCm
        if backend == 'nccl':
            torch.distributed.init_process_group('nccl', device_id=torch.device('cuda', local))
        else:
            torch.distributed.init_process_group(backend)
    return torch.distributed.get_rank(), world


def _agree_on_output_path(args, rank, world):
    """handle_output_path on rank 0; its verdict (0 or the exit code) is broadcast so that every rank leaves together
    instead of the others hanging in the first collective until the RCCL timeout."""
    cache_path, code = None, 0
    if rank == 0:
        try:
            cache_path = handle_output_path(args)
        except SystemExit as e:
            code = int(e.code) if isinstance(e.code, int) else 1
    grouped = world > 1 or torch.distributed.is_initialized()       # (a forced one-rank group takes the same calls)
    if grouped:
        dev = torch.device('cuda', torch.cuda.current_device()) if torch.distributed.get_backend() == 'nccl' else 'cpu'
        verdict = torch.tensor([code], dtype=torch.int32, device=dev)
        torch.distributed.broadcast(verdict, src=0)
        code = int(verdict.item())
    if code != 0:
        if grouped:
            torch.distributed.destroy_process_group()
        sys.exit(code)
    return cache_path


def main(argv=None):
    from argparse import ArgumentParser
    dino_archs = ['vits16', 'vits8', 'vitb16', 'vitb8']
    dino2_archs = ['vits14', 'vitb14', 'vitl14', 'vitg14']
    parser = ArgumentParser('Infer DINO features from saved volume')
    parser.add_argument('--data-path', type=str, required=True, help='volume file (.npy / .pt) to extract features from')
    parser.add_argument('--cache-path', type=str, default=None, help='where the feature file goes (default: next to the volume)')
    parser.add_argument('--dino-model', type=str, choices=dino_archs, default=None, help='DINO ViT variant')
    parser.add_argument('--dino2-model', type=str, choices=dino2_archs, default=None, help='DINOv2 variant (not available offline)')
    parser.add_argument('--slice-along', type=str, choices=['x', 'y', 'z', 'all'], default='all',
                        help='Along which axis to slice volume, as it is fed slice-wise to DINO')
    parser.add_argument('--batch-size', type=int, default=1, help='a LOWER bound on the slices per engine call (the engine sizes its own calls: 256 x 4097 / tokens by '
                        'default); the memory knob is $VITTF_ENGINE_BATCH')
    parser.add_argument('--feature-output-size', type=int, default=64,
                        help='Produces a features map with aspect ratio of input volume with this value as y resolution. Only if --slice-along ALL')
    parser.add_argument('--cpu', action='store_true', help='Use CPU only (not available in the MI355X build)')
    parser.add_argument('--overwrite', action='store_true', help='replace an existing feature file')
    # additions of the MI355X build
    parser.add_argument('--weights', type=str, default=None, help='Local DINO state dict (.pth)')
    parser.add_argument('--synthetic-weights', type=int, default=None, metavar='SEED', help='Use seeded synthetic weights')
    parser.add_argument('--engine-dtype', type=str, choices=['fp16', 'bf16'], default='fp16',
                        help='MFMA operand type (fp16 = the reference GPU autocast type, 1e-3 parity; bf16 opt-in)')
    parser.add_argument('--attention', type=str, choices=['16bit', 'fp8'], default='16bit',
                        help="fp8: e4m3 attention on the block-scaled matrix instruction (BASELINE configs[3]; ~3e-2 on the features)")
    args = parser.parse_args(argv)

    if args.cpu:
        print('--cpu: this build has no CPU path (the hot path runs only on MI355X / gfx950).')
        sys.exit(1)
    _MODEL_OPTS.update(weights=args.weights, synthetic_seed=args.synthetic_weights, dtype=args.engine_dtype, attention=args.attention)
    dino_model, dino_model_fn, patch_size = load_model(args)
    rank, world = _init_distributed()
    cache_path = _agree_on_output_path(args, rank, world)

    vol = load_data(args.data_path, keep_dtype=True)
    im_sz, feat_out_sz = vt.sizing(tuple(vol.shape), args.feature_output_size, patch_size)
    print(f'Input image size: {im_sz}')
    model = dino_model_fn(dino_model)
    torch.cuda.synchronize()
    t0 = time.time()
    eb = vt.extract.AtLeast(args.batch_size) if args.batch_size > 1 else None   # a lower bound (extract.engine_batch_for)
    feats = vt.feature_volume(vol, model, args.feature_output_size, args.slice_along, eb)
    if args.slice_along == 'all':
        qkv = defaultdict(float)           # the reference saves a defaultdict in 'all' mode (:328)
        qkv['k'] = feats.cpu()
    else:
        qkv = {'k': feats.cpu()}
    if rank == 0:
        print('k', ':', qkv['k'].shape)
        print(f'Computed qkv along {args.slice_along} in {time.time() - t0}s, saving now to: {cache_path}')
        save_features(qkv, cache_path)
    if world > 1 or torch.distributed.is_initialized():
        if rank == 0:
            print('slab exchanges:', ', '.join(f'{n} over {b}' for b, n in sorted(vt.extract.EXCHANGES.items())) or 'none')
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()
    sys.exit(0)


if __name__ == '__main__':
    main()
