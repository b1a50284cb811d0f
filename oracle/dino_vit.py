"""Oracle: DINO VisionTransformer forward, restated for the CPU (fp32).

TEST INFRASTRUCTURE -- see oracle/__init__.py.

The reference obtains this model at run time with
``torch.hub.load('facebookresearch/dino:main', f'dino_{name}')``
(infer.py:42-43, called at infer.py:323).  That repository is not vendored and
there is no network, so the published algorithm (Caron et al. 2021, DINO
``vision_transformer.py``; hub entries ``dino_vits8`` = ``vit_small(patch_size=8,
num_classes=0)``, ``dino_vitb8`` = ``vit_base(patch_size=8, num_classes=0)``)
is restated here.  The module tree and parameter names follow the DINO
state-dict layout so that (a) real DINO checkpoints load with
``load_state_dict`` from a local file and (b) the reference harness can hook
``model._modules["blocks"][-1]._modules["attn"]._modules["qkv"]`` and read
``.attn.num_heads`` exactly as it does upstream (infer.py:135, infer.py:180).

Details that matter for parity:
  * PatchEmbed = Conv2d(3 -> D, kernel P, stride P), flatten(2).transpose(1, 2)
  * CLS token prepended, then the position embedding is added; for an image
    whose patch grid differs from the stored 28x28 grid the patch part of the
    embedding is resized with *bicubic* interpolation in the scale-factor form
    ``scale_factor=((w0 + 0.1) / sqrt(N), (h0 + 0.1) / sqrt(N))``
  * pre-norm blocks, LayerNorm eps = 1e-6, qkv Linear with bias,
    softmax(q k^T / sqrt(d_h)) v, exact (erf) GELU, no dropout in eval
  * forward returns the CLS row of the final norm (discarded by the reference,
    infer.py:177 ``_ = model(...)``).
"""
import math
from functools import partial

import torch
import torch.nn as nn
import torch.nn.functional as F

ARCHS = {
    # name: (embed_dim, depth, heads, patch)
    'vits8': (384, 12, 6, 8),
    'vits16': (384, 12, 6, 16),
    'vitb8': (768, 12, 12, 8),
    'vitb16': (768, 12, 12, 16),
}


class Mlp(nn.Module):
    def __init__(self, dim, hidden):
        super().__init__()
        self.fc1 = nn.Linear(dim, hidden)
        self.act = nn.GELU()           # erf form, not tanh
        self.fc2 = nn.Linear(hidden, dim)

    def forward(self, x):
        return self.fc2(self.act(self.fc1(x)))


class Attention(nn.Module):
    def __init__(self, dim, num_heads):
        super().__init__()
        self.num_heads = num_heads
        self.scale = (dim // num_heads) ** -0.5
        self.qkv = nn.Linear(dim, 3 * dim, bias=True)
        self.proj = nn.Linear(dim, dim)

    def forward(self, x):
        b, n, c = x.shape
        # (3, B, heads, N, d_h): output columns of qkv are [q | k | v], heads
        # concatenated inside each third -- this is what infer.py:189-193 undoes
        qkv = self.qkv(x).reshape(b, n, 3, self.num_heads, c // self.num_heads).permute(2, 0, 3, 1, 4)
        q, k, v = qkv[0], qkv[1], qkv[2]
        att = (q @ k.transpose(-2, -1)) * self.scale
        att = att.softmax(dim=-1)
        y = (att @ v).transpose(1, 2).reshape(b, n, c)
        return self.proj(y)


class Block(nn.Module):
    def __init__(self, dim, num_heads, mlp_ratio=4.0, eps=1e-6):
        super().__init__()
        self.norm1 = nn.LayerNorm(dim, eps=eps)
        self.attn = Attention(dim, num_heads)
        self.norm2 = nn.LayerNorm(dim, eps=eps)
        self.mlp = Mlp(dim, int(dim * mlp_ratio))

    def forward(self, x):
        x = x + self.attn(self.norm1(x))
        x = x + self.mlp(self.norm2(x))
        return x


class PatchEmbed(nn.Module):
    def __init__(self, patch_size, embed_dim, in_chans=3):
        super().__init__()
        self.patch_size = patch_size
        self.proj = nn.Conv2d(in_chans, embed_dim, kernel_size=patch_size, stride=patch_size)

    def forward(self, x):
        return self.proj(x).flatten(2).transpose(1, 2)


def interpolate_pos_embed(pos_embed, npatch, rows, cols, patch_size):
    """Position embedding for an image of ``rows x cols`` pixels.

    pos_embed: (1, 1 + G*G, D) stored embedding (G = 28 for the 224/8 models).
    Returns (1, 1 + npatch, D).  Bicubic, scale-factor form with the +0.1
    fudge of the upstream implementation; identity when the grid already
    matches and the image is square.
    """
    n_stored = pos_embed.shape[1] - 1
    if npatch == n_stored and rows == cols:
        return pos_embed
    cls_pos = pos_embed[:, :1]
    grid_pos = pos_embed[:, 1:]
    dim = pos_embed.shape[-1]
    g = int(math.sqrt(n_stored))
    r0 = rows // patch_size + 0.1
    c0 = cols // patch_size + 0.1
    grid_pos = F.interpolate(
        grid_pos.reshape(1, g, g, dim).permute(0, 3, 1, 2),
        scale_factor=(r0 / math.sqrt(n_stored), c0 / math.sqrt(n_stored)),
        mode='bicubic')
    assert int(r0) == grid_pos.shape[-2] and int(c0) == grid_pos.shape[-1]
    grid_pos = grid_pos.permute(0, 2, 3, 1).reshape(1, -1, dim)
    return torch.cat((cls_pos, grid_pos), dim=1)


class VisionTransformer(nn.Module):
    def __init__(self, patch_size=8, embed_dim=384, depth=12, num_heads=6, mlp_ratio=4.0,
                 stored_img_size=224):
        super().__init__()
        self.embed_dim = embed_dim
        self.patch_embed = PatchEmbed(patch_size, embed_dim)
        n_stored = (stored_img_size // patch_size) ** 2
        self.cls_token = nn.Parameter(torch.zeros(1, 1, embed_dim))
        self.pos_embed = nn.Parameter(torch.zeros(1, n_stored + 1, embed_dim))
        self.blocks = nn.ModuleList([Block(embed_dim, num_heads, mlp_ratio) for _ in range(depth)])
        self.norm = nn.LayerNorm(embed_dim, eps=1e-6)

    def prepare_tokens(self, x):
        b, _, rows, cols = x.shape
        tok = self.patch_embed(x)
        tok = torch.cat((self.cls_token.expand(b, -1, -1), tok), dim=1)
        return tok + interpolate_pos_embed(self.pos_embed, tok.shape[1] - 1, rows, cols,
                                           self.patch_embed.patch_size)

    def forward(self, x):
        x = self.prepare_tokens(x)
        for blk in self.blocks:
            x = blk(x)
        return self.norm(x)[:, 0]

    # ---- helpers used by tests (not part of the upstream surface) ----
    def tokens_before_block(self, x, idx):
        """Residual stream entering block ``idx`` (0-based)."""
        x = self.prepare_tokens(x)
        for blk in self.blocks[:idx]:
            x = blk(x)
        return x

    def last_block_k(self, x):
        """K third of blocks[-1].attn.qkv for every token (B, N, D), fp32."""
        t = self.tokens_before_block(x, len(self.blocks) - 1)
        blk = self.blocks[-1]
        d = self.embed_dim
        return F.linear(blk.norm1(t), blk.attn.qkv.weight[d:2 * d], blk.attn.qkv.bias[d:2 * d])


def build_vit(arch='vits8', state_dict=None, **overrides):
    """Construct the oracle model; ``arch`` is a DINO name or a (D, depth, heads, patch) tuple."""
    dim, depth, heads, patch = ARCHS[arch] if isinstance(arch, str) else arch
    model = VisionTransformer(patch_size=patch, embed_dim=dim, depth=depth, num_heads=heads, **overrides)
    if state_dict is not None:
        model.load_state_dict(state_dict, strict=True)
    return model.eval()
