"""Weights of the DINO ViT in the upstream state-dict layout: local loading, a seeded synthetic
recipe, and the host-side preprocessing the HIP engine needs (channel folding, position embedding).

The reference fetches the model over the network (``torch.hub.load('facebookresearch/dino:main',
'dino_vits8')``, infer.py:42-43).  Here weights come from a LOCAL state-dict file (same key names as the
DINO checkpoints: ``cls_token``, ``pos_embed``, ``patch_embed.proj.*``, ``blocks.{i}.norm1|attn.qkv|
attn.proj|norm2|mlp.fc1|mlp.fc2.*``, ``norm.*``) or from the seeded synthetic recipe below when no
checkpoint is available (benchmarks, tests).
"""
import math
import os

import torch
import torch.nn.functional as F

# name: (embed_dim, depth, heads, patch)   -- hub entries dino_<name>
ARCHS = {
    'vits8': (384, 12, 6, 8),
    'vits16': (384, 12, 6, 16),
    'vitb8': (768, 12, 12, 8),
    'vitb16': (768, 12, 12, 16),
}
# file names torch.hub would have cached for these entries
HUB_FILES = {
    'vits8': 'dino_deitsmall8_pretrain.pth',
    'vits16': 'dino_deitsmall16_pretrain.pth',
    'vitb8': 'dino_vitbase8_pretrain.pth',
    'vitb16': 'dino_vitbase16_pretrain.pth',
}
IN_MEAN = (0.485, 0.456, 0.406)   # infer.py:39
IN_STD = (0.229, 0.224, 0.225)    # infer.py:40


def arch_of(arch):
    if isinstance(arch, str):
        if arch not in ARCHS:
            raise ValueError(f'unknown DINO arch {arch!r}; known: {sorted(ARCHS)}')
        return ARCHS[arch]
    dim, depth, heads, patch = arch
    return int(dim), int(depth), int(heads), int(patch)


# "massive activation" channels of the outlier variant below (trained ViTs carry a handful of residual-stream channels two
# orders of magnitude above the rest; Gaussian unit-gain weights have none)
OUTLIER_CHANNELS = (7, 100, 191, 250, 333, 380)


def synthetic_state_dict(arch='vits8', seed=0, stored_grid=28, outliers=False):
    """Seeded random weights with the DINO key layout.

    outliers=True: the same weights with six massive channels planted -- x50 rows in two blocks' mlp.fc2 and one block's
    attn.proj (the residual stream then carries them to the end), x50 entries in several norm weights (16-bit LayerNorm
    output in the hundreds), and in one block a q / k pair of one head that both read the same massive direction, so that
    its attention logits spread over more than +-60: what the parity tests on benign weights never drive through the 16-bit
    operand path (LN -> h, GELU -> hidden, the lazy-maximum attention kernel's overflow branch).

    Not an initialisation for training: the scales are chosen so that a forward pass looks like a trained
    ViT numerically (unit-gain linears, attention logits with a std of a few units so the softmax is
    peaked, non-trivial biases and LayerNorm affine terms) -- this keeps parity tests sensitive to
    ordering / bias / masking mistakes and gives realistic data for benchmarking.
    """
    dim, depth, heads, patch = arch_of(arch)
    g = torch.Generator().manual_seed(seed)

    def rnd(*shape, std=1.0):
        return torch.randn(*shape, generator=g) * std

    sd = {
        'cls_token': rnd(1, 1, dim, std=0.5),
        'pos_embed': rnd(1, stored_grid * stored_grid + 1, dim, std=0.5),
        'patch_embed.proj.weight': rnd(dim, 3, patch, patch, std=1.0 / math.sqrt(3 * patch * patch) * 3.0),
        'patch_embed.proj.bias': rnd(dim, std=0.1),
        'norm.weight': 1.0 + rnd(dim, std=0.1),
        'norm.bias': rnd(dim, std=0.1),
    }
    for i in range(depth):
        p = f'blocks.{i}.'
        sd[p + 'norm1.weight'] = 1.0 + rnd(dim, std=0.1)
        sd[p + 'norm1.bias'] = rnd(dim, std=0.1)
        sd[p + 'attn.qkv.weight'] = rnd(3 * dim, dim, std=1.3 / math.sqrt(dim))
        sd[p + 'attn.qkv.bias'] = rnd(3 * dim, std=0.1)
        sd[p + 'attn.proj.weight'] = rnd(dim, dim, std=1.0 / math.sqrt(dim))
        sd[p + 'attn.proj.bias'] = rnd(dim, std=0.1)
        sd[p + 'norm2.weight'] = 1.0 + rnd(dim, std=0.1)
        sd[p + 'norm2.bias'] = rnd(dim, std=0.1)
        sd[p + 'mlp.fc1.weight'] = rnd(4 * dim, dim, std=1.0 / math.sqrt(dim))
        sd[p + 'mlp.fc1.bias'] = rnd(4 * dim, std=0.1)
        sd[p + 'mlp.fc2.weight'] = rnd(dim, 4 * dim, std=1.0 / math.sqrt(4 * dim))
        sd[p + 'mlp.fc2.bias'] = rnd(dim, std=0.1)
    if outliers:
        c = [ch % dim for ch in OUTLIER_CHANNELS]
        for blk in (1, depth // 2):
            sd[f'blocks.{blk}.mlp.fc2.weight'][c[0]] *= 50.0
            sd[f'blocks.{blk}.mlp.fc2.weight'][c[1]] *= 50.0
        sd[f'blocks.{2 % depth}.attn.proj.weight'][c[2]] *= 50.0
        for blk in range(depth):
            sd[f'blocks.{blk}.norm1.weight'][c[3]] *= 50.0 if blk % 3 == 0 else 1.0
            sd[f'blocks.{blk}.norm2.weight'][c[4]] *= 50.0 if blk % 3 == 1 else 1.0
        # one head whose q and k both follow the (massive) channel c[0] of the normalised input: logits = 64 a^2 h_q h_k / 8
        # (all of one sign there: logits up to ~140 but a spread of only ~10 inside a row), and one head that follows a
        # channel whose sign changes with the token's position (its position embedding is x 20): logits of both signs (+-60)
        # inside one row, i.e. keys far above the first key tile's maximum late in the sequence
        sd['pos_embed'][0, :, c[5]] *= 20.0
        blk = min(depth - 2, 6)
        wq = sd[f'blocks.{blk}.attn.qkv.weight']
        for head, ch, a in ((1 % heads, c[0], 0.5), (2 % heads, c[5], 0.5)):
            wq[head * 64:(head + 1) * 64, ch] += a
            wq[dim + head * 64:dim + (head + 1) * 64, ch] += a
    return sd


def load_state_dict_file(path):
    """Load a DINO checkpoint from a local file; accepts bare backbones and teacher/student wrappers."""
    sd = torch.load(path, map_location='cpu', weights_only=True)
    for key in ('teacher', 'student', 'state_dict', 'model'):
        if isinstance(sd, dict) and key in sd and isinstance(sd[key], dict):
            sd = sd[key]
            break
    out = {}
    for k, v in sd.items():
        for prefix in ('module.', 'backbone.'):
            if k.startswith(prefix):
                k = k[len(prefix):]
        if k.startswith('head.'):
            continue
        out[k] = v.float()
    return out


def find_local_checkpoint(name):
    """Where a previously downloaded hub checkpoint would be; None if absent.  Never touches the network."""
    env = os.environ.get('VITTF_WEIGHTS')
    if env:
        return env if os.path.exists(env) else None
    hub = os.environ.get('TORCH_HOME', os.path.join(os.path.expanduser('~'), '.cache', 'torch'))
    cand = os.path.join(hub, 'hub', 'checkpoints', HUB_FILES.get(name, ''))
    return cand if os.path.isfile(cand) else None


def state_dict_checksum(sd):
    """Order-independent float64 checksum used by golden fixtures to detect generator drift."""
    tot = 0.0
    for k in sorted(sd):
        t = sd[k].double()
        tot += float((t * torch.arange(1, t.numel() + 1, dtype=torch.float64).reshape(t.shape).remainder(7.0)).sum())
    return tot


def fold_patch_embed(weight, bias):
    """Fold the 3-channel conv applied to a grey image replicated over 3 ImageNet-normalised channels
    into a single-channel conv acting on the [0, 1] grey value:

        sum_c W[d,c,i,j] * (x - mean_c) / std_c + b[d]
      = sum_ij (sum_c W[d,c,i,j] / std_c) * x_ij + (b[d] - sum_c mean_c / std_c * sum_ij W[d,c,i,j])

    (infer.py:39-40, 154-155 + upstream PatchEmbed).  Returns (w_t [P*P][D], b [D]) fp32, folded in fp64.
    """
    w = weight.double()
    mean = torch.tensor(IN_MEAN, dtype=torch.float64).view(1, 3, 1, 1)
    std = torch.tensor(IN_STD, dtype=torch.float64).view(1, 3, 1, 1)
    w1 = (w / std).sum(1)                                  # (D, P, P)
    b1 = bias.double() - (w * mean / std).sum((1, 2, 3))
    d = w.shape[0]
    return w1.reshape(d, -1).t().contiguous().float(), b1.float()


def interpolate_pos_embed(pos_embed, rows, cols, patch):
    """Position embedding for a rows x cols pixel image, upstream form: bicubic,
    ``scale_factor=((r0 + 0.1) / sqrt(N), (c0 + 0.1) / sqrt(N))``; identity for the stored square grid.
    Runs once per image size on the host (it depends on the weights only).  Returns (1, 1 + r0*c0, D)."""
    n_stored = pos_embed.shape[1] - 1
    r0, c0 = rows // patch, cols // patch
    if r0 * c0 == n_stored and rows == cols:
        return pos_embed
    dim = pos_embed.shape[-1]
    g = int(math.sqrt(n_stored))
    grid = pos_embed[:, 1:].reshape(1, g, g, dim).permute(0, 3, 1, 2)
    grid = F.interpolate(grid.float(), scale_factor=((r0 + 0.1) / math.sqrt(n_stored), (c0 + 0.1) / math.sqrt(n_stored)),
                         mode='bicubic')
    if grid.shape[-2] != r0 or grid.shape[-1] != c0:
        raise ValueError(f'position-embedding grid {tuple(grid.shape[-2:])} != token grid {(r0, c0)}')
    grid = grid.permute(0, 2, 3, 1).reshape(1, -1, dim)
    return torch.cat((pos_embed[:, :1].float(), grid), dim=1)


FC2_PERM16 = (0, 1, 2, 3, 8, 9, 10, 11, 4, 5, 6, 7, 12, 13, 14, 15)


def _tile_pos_tables():
    """(row, chunk) held by each of the 256 16-byte positions of a [32 rows][64 x 16-bit] sub-image in the kernels' tile_off
    layout (vittf_common.h: tile_pos)."""
    q = torch.arange(256)
    p = q >> 4
    slot = (q & 15) ^ (p & 15)
    return (p << 1) | (slot >> 3), slot & 7


def _pack_row_images(w):
    """[L, 32 U, 384] -> [L, U, 12288]: image u = rows 32 u .. + 31 over the 384 inputs as 6 sub-images [32 rows][64 k] in LDS
    layout: element (s6, q, e) = w[32 u + r[q], 64 s6 + 8 c[q] + e]."""
    L, n, d = w.shape
    assert d == 384 and n % 32 == 0
    units, dev = n // 32, w.device
    r, c = (t.to(dev) for t in _tile_pos_tables())
    e = torch.arange(8, device=dev)
    s6 = torch.arange(6, device=dev)
    rows = torch.arange(units, device=dev).view(-1, 1, 1, 1) * 32 + r.view(1, 1, -1, 1)                # [U, 1, 256, 1]
    cols = (64 * s6.view(1, -1, 1, 1) + 8 * c.view(1, 1, -1, 1) + e.view(1, 1, 1, -1))                  # [1, 6, 256, 8]
    return w[:, rows.expand(units, 6, 256, 8), cols.expand(units, 6, 256, 8)].reshape(L, units, -1)


def pack_row_images(w):
    """A K = 384 weight [L, N, 384] (N a multiple of 32) as the stream of 24 KB LDS images of vittf_gemm_as (csrc/gemm_as.hip):
    image u = rows 32 u .. + 31, natural k order -> [L, N / 32, 12288]."""
    return _pack_row_images(w).contiguous()


def norm2_register_order():
    """Column of x held at fc1 input position 16 s + 8 h + e when norm2 is computed in the block-tail kernel's accumulator
    registers (csrc/mlp.hip): 32 (s >> 1) + 16 (s & 1) + 8 (e >> 2) + 4 h + (e & 3)."""
    p = torch.arange(384)
    s_, h, e = p >> 4, (p >> 3) & 1, p & 7
    return 32 * (s_ >> 1) + 16 * (s_ & 1) + 8 * (e >> 2) + 4 * h + (e & 3)


def _mlp_images(w1, w2):
    """fc1 / fc2 weights of L blocks as 24 KB LDS images, one per hidden unit of 32: (img1, img2), each [L, 48, 12288].
      W1(u): rows = hidden units 32 u .. + 31, k = the 384 inputs: 6 sub-images [32 rows][64 k] (tile_off layout).
      W2(u): rows = outputs; sub-image s6 holds output tiles 2 s6 and 2 s6 + 1, 32 k each = hidden units 32 u .. + 31 in the
             order the second MFMA finds them in the first one's accumulator registers (permute_fc2_hidden)."""
    L, hid, d = w1.shape
    assert d == 384 and hid == 4 * d and tuple(w2.shape) == (L, d, hid)
    units = hid // 32
    dev = w1.device
    r, c = (t.to(dev) for t in _tile_pos_tables())                      # [256]
    e = torch.arange(8, device=dev)
    s6 = torch.arange(6, device=dev)
    img1 = _pack_row_images(w1)
    # W2(u): element (s6, q, e) = w2p[32 (2 s6 + (c >> 2)) + r, 32 u + 16 ((c >> 1) & 1) + 8 (c & 1) + e]
    w2p = permute_fc2_hidden(w2)
    rows2 = (32 * (2 * s6.view(1, -1, 1, 1) + (c.view(1, 1, -1, 1) >> 2)) + r.view(1, 1, -1, 1))       # [1, 6, 256, 1]
    cols2 = (32 * torch.arange(units, device=dev).view(-1, 1, 1, 1) + 16 * ((c.view(1, 1, -1, 1) >> 1) & 1)
             + 8 * (c.view(1, 1, -1, 1) & 1) + e.view(1, 1, 1, -1))                                     # [U, 1, 256, 8]
    img2 = w2p[:, rows2.expand(units, 6, 256, 8), cols2.expand(units, 6, 256, 8)].reshape(L, units, -1)
    return img1, img2


TAIL_FX_PSTEPS, TAIL_FX_MSTEPS, TAIL_FX_LAG = 12, 100, 4


def _pack_proj_kmajor(wp):
    """[L, 384, 384] -> [L, 12, 12288]: projection step p = k steps 2 p, 2 p + 1 of all 12 output tiles: fragment f (0 .. 23; the
    kernel reads it at sub-image f >> 2, chunk pair f & 3) = Wp[32 (f % 12) + r][16 (2 p + f // 12) + 8 h + e]."""
    L, n, d = wp.shape
    assert n == 384 and d == 384
    dev = wp.device
    r, c = (t.to(dev) for t in _tile_pos_tables())
    e = torch.arange(8, device=dev).view(1, 1, 1, -1)
    s6 = torch.arange(6, device=dev).view(1, -1, 1, 1)
    p = torch.arange(12, device=dev).view(-1, 1, 1, 1)
    f = 4 * s6 + (c.view(1, 1, -1, 1) >> 1)                                 # [1, 6, 256, 1]
    rows = 32 * (f % 12) + r.view(1, 1, -1, 1)
    cols = 16 * (2 * p + f // 12) + 8 * (c.view(1, 1, -1, 1) & 1) + e        # [12, 6, 256, 8]
    return wp[:, rows.expand(12, 6, 256, 8), cols.expand(12, 6, 256, 8)].reshape(L, 12, -1)


def pack_block_tail_weights(wp, w1, w2):
    """proj / fc1 / fc2 weights of L blocks -> the stream of the block-tail kernel (vittf_block_tail, csrc/tail_fx.hip): per block
    12 projection steps of 24 KB (k steps 2 p, 2 p + 1 of all output tiles, K-major: _pack_proj_kmajor), then 100 main steps =
    [ W1(m >> 1) k half m & 1 | W2((m - 4) >> 1) output-tile half (m - 4) & 1 ] (zeros where a role has no work: the first four
    W2 halves, the last four W1 halves); fc1's input dim in the order norm2 leaves its values in the registers.
    -> [L, 112, 12288]."""
    L, d, d2 = wp.shape
    assert d == 384 and d2 == 384
    order = norm2_register_order().to(w1.device)
    img1, img2 = _mlp_images(w1[:, :, order], w2)
    half = img1.shape[-1] // 2
    zero = torch.zeros_like(img1[:, 0, :half])
    steps = []
    for m in range(TAIL_FX_MSTEPS):
        a = img1[:, m >> 1, (m & 1) * half:(m & 1) * half + half] if m < 96 else zero
        m2 = m - TAIL_FX_LAG
        b = img2[:, m2 >> 1, (m2 & 1) * half:(m2 & 1) * half + half] if m2 >= 0 else zero
        steps.append(torch.cat([a, b], dim=-1))
    return torch.cat([_pack_proj_kmajor(wp), torch.stack(steps, dim=1)], dim=1).contiguous()


def permute_fc2_hidden(w2):
    """fc2 weight [..., D, 4D] with its hidden (input) dim re-ordered inside every block of 16: the k order in which
    the fused MLP kernel's second MFMA consumes the first one's accumulator registers (include/vittf.h, fc2_w_perm)."""
    hid = w2.shape[-1]
    idx = (torch.arange(hid).view(-1, 16)[:, list(FC2_PERM16)]).reshape(-1)
    return w2[..., idx].contiguous()
