"""HipViT: the DINO ViT as an HBM-resident weight set driven through libvittf's C ABI.

Stands where ``torch.hub.load('facebookresearch/dino:main', 'dino_vits8').to(dev).eval()`` stands in
the reference (infer.py:323): an object the extraction loop hands slices to.  It exposes what the
reference reads off the upstream module (``num_heads`` via ``blocks[-1].attn.num_heads`` infer.py:180,
the patch size, the embedding width) but runs nothing in PyTorch: torch only owns the device buffers.
"""
import ctypes as C

import torch

from . import _lib
from .weights import arch_of, fold_patch_embed, interpolate_pos_embed, pack_block_tail_weights, pack_row_images

_TORCH_DT = {_lib.BF16: torch.bfloat16, _lib.FP16: torch.float16}


class _Attn:
    def __init__(self, heads):
        self.num_heads = heads


class _Block:
    def __init__(self, heads):
        self.attn = _Attn(heads)


class HipViT:
    """Weights + workspace of one ViT on one GPU.

    state_dict: DINO-layout tensors (fp32, CPU or GPU).  arch: DINO name ('vits8', ...) or
    (embed_dim, depth, heads, patch).  dtype of the MFMA operands: 'fp16' (default: the reference's own GPU autocast
    type, infer.py:309; meets the 1e-3 parity bound against the fp32 CPU path) or 'bf16' (opt-in: 8-bit mantissa,
    2.4e-3 .. 3.9e-3 against the CPU path).  attention: '16bit' (default) or 'fp8' -- BASELINE configs[3]'s fp8 MFMA
    attention path (e4m3 operands on the block-scaled matrix instruction; ~3e-2 on the features: opt-in).
    fused_tail=False: ViT-S without the packed weight streams (the block tail and the activation-stationary qkv GEMM): the GEMM
    launches every other width uses; flags: _lib.CFG_* bits (vittf_vit_config.flags), the slower alternatives the tests also run.
    """

    def __init__(self, state_dict, arch='vits8', dtype='fp16', device=None, attention='16bit', fused_tail=True, flags=0):
        self.lib = _lib.require_device()
        self.device = torch.device(device if device is not None else f'cuda:{torch.cuda.current_device()}')
        dim, depth, heads, patch = arch_of(arch)
        if heads * 64 != dim:
            raise ValueError('HipViT supports head dim 64 only (all DINO ViTs)')
        self.embed_dim, self.depth, self.num_heads, self.patch_size = dim, depth, heads, patch
        self.dtype_id = _lib.DTYPES[dtype] if isinstance(dtype, str) else int(dtype)
        self.dtype_name = 'bf16' if self.dtype_id == _lib.BF16 else 'fp16'
        self.blocks = [_Block(heads) for _ in range(depth)]    # duck-typing of model.blocks[-1].attn.num_heads
        if attention not in ('16bit', 'fp8'):
            raise ValueError(f"attention must be '16bit' or 'fp8', got {attention!r}")
        self.attention = attention
        self.cfg = _lib.VitConfig(dim, depth, heads, patch, self.dtype_id, 1e-6, 1 if attention == 'fp8' else 0, int(flags))

        sd = {k: v.detach().float().cpu() for k, v in state_dict.items()}
        h16 = _TORCH_DT[self.dtype_id]
        dev = self.device

        def stack(fmt, dt):
            return torch.stack([sd[fmt.format(i)] for i in range(depth)]).to(dev, dt).contiguous()

        pe_w_t, pe_b = fold_patch_embed(sd['patch_embed.proj.weight'], sd['patch_embed.proj.bias'])
        self._t = {
            'pe_w_t': pe_w_t.to(dev).contiguous(), 'pe_b': pe_b.to(dev).contiguous(),
            'qkv_w': stack('blocks.{}.attn.qkv.weight', h16), 'qkv_b': stack('blocks.{}.attn.qkv.bias', torch.float32),
            'proj_w': stack('blocks.{}.attn.proj.weight', h16), 'proj_b': stack('blocks.{}.attn.proj.bias', torch.float32),
            'fc1_w': stack('blocks.{}.mlp.fc1.weight', h16), 'fc1_b': stack('blocks.{}.mlp.fc1.bias', torch.float32),
            'fc2_w': stack('blocks.{}.mlp.fc2.weight', h16), 'fc2_b': stack('blocks.{}.mlp.fc2.bias', torch.float32),
            'ln1_g': stack('blocks.{}.norm1.weight', torch.float32), 'ln1_b': stack('blocks.{}.norm1.bias', torch.float32),
            'ln2_g': stack('blocks.{}.norm2.weight', torch.float32), 'ln2_b': stack('blocks.{}.norm2.bias', torch.float32),
        }
        ptrs = {k: v.data_ptr() for k, v in self._t.items()}
        ptrs['tail_packed'] = ptrs['qkv_packed'] = None
        if fused_tail and dim == 384:          # the kernels' register budgets are sized for ViT-S
            self._t['tail_packed'] = pack_block_tail_weights(self._t['proj_w'], self._t['fc1_w'], self._t['fc2_w'])
            self._t['qkv_packed'] = pack_row_images(self._t['qkv_w'])
            ptrs['tail_packed'] = self._t['tail_packed'].data_ptr()
            ptrs['qkv_packed'] = self._t['qkv_packed'].data_ptr()
        self.weights = _lib.VitWeights(**ptrs)
        self._cls = sd['cls_token'].reshape(1, 1, dim)
        self._pos = sd['pos_embed']
        self._pos_cache = {}
        self._ws = None

    # -- reference-shaped conveniences --------------------------------------------------------------
    def to(self, *_a, **_k):
        return self

    def eval(self):
        return self

    # -- device-side pieces ---------------------------------------------------------------------------
    def pos_for(self, rows, cols):
        """Device position embedding for a rows x cols image: (PosEmbed struct, keep-alive tensors)."""
        key = (rows, cols)
        if key not in self._pos_cache:
            pos = interpolate_pos_embed(self._pos, rows, cols, self.patch_size)[0]      # (1 + n, D)
            cls0 = (self._cls[0, 0] + pos[0]).to(self.device).contiguous()
            patch = pos[1:].to(self.device).contiguous()
            self._pos_cache[key] = (_lib.PosEmbed(cls0.data_ptr(), patch.data_ptr()), cls0, patch)
        return self._pos_cache[key]

    def workspace(self, batch, tokens):
        """The engine's workspace for `batch` slices of `tokens` tokens (grown on demand, kept)."""
        need = self.lib.vittf_vit_workspace_bytes(C.byref(self.cfg), batch, tokens)
        if need == 0:
            raise _lib.VittfError('unsupported ViT configuration for the HIP engine')
        if self._ws is None or self._ws.numel() < need:
            self._ws = None
            self._ws = torch.empty(need, dtype=torch.uint8, device=self.device)
        return self._ws

    def k_features(self, view, slice0, batch, out, part=1):
        """Run slices [slice0, slice0+batch) of `view` (a _lib.SliceView) through the ViT and write the
        hooked qkv third (`part`: 0 q, 1 k, 2 v) of the patch tokens as fp16 into `out`
        (tensor of >= batch * f0*f1 * D halves)."""
        p = self.patch_size
        tokens = (view.out_rows // p) * (view.out_cols // p) + 1
        pos, _, _ = self.pos_for(view.out_rows, view.out_cols)
        ws = self.workspace(batch, tokens)
        assert out.dtype == torch.float16 and out.is_contiguous() and out.numel() >= batch * (tokens - 1) * self.embed_dim
        rc = self.lib.vittf_vit_k_features(C.byref(self.cfg), C.byref(self.weights), C.byref(pos), C.byref(view),
                                           slice0, batch, part, _lib.ptr(out), _lib.ptr(ws), ws.numel(),
                                           _lib.stream_ptr())
        _lib.check(rc, 'vittf_vit_k_features')

    def __call__(self, *_a, **_k):
        raise _lib.VittfError('HipViT is driven through compute_qkv / FeatureExtractor, not called on image tensors')
