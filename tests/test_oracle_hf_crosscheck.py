"""CPU: oracle/dino_vit.py against an independently written ViT that ships in the image (transformers.ViTModel built
from a config object -- no download), weights mapped key by key, 224 x 224 input (no position-embedding resize).

This does NOT pin the oracle to upstream facebookresearch/dino (absent from /root/reference and from the image: row a5
stays "parity unpinned"); it removes the single-author risk on the block arithmetic: pre-norm residual blocks, LayerNorm
eps 1e-6, qkv bias, the [q | k | v] column order with heads concatenated inside each third, softmax(q k^T / 8) v, the
erf GELU and the final norm all have to agree for the numbers below to match.
"""
import pytest
import torch

import vit_tf_amd as vt
from oracle import dino_vit

transformers = pytest.importorskip('transformers')


def _hf_from_dino(sd, dim, depth, heads, patch):
    cfg = transformers.ViTConfig(hidden_size=dim, num_hidden_layers=depth, num_attention_heads=heads,
                                 intermediate_size=4 * dim, hidden_act='gelu', hidden_dropout_prob=0.0,
                                 attention_probs_dropout_prob=0.0, layer_norm_eps=1e-6, image_size=224, patch_size=patch,
                                 num_channels=3, qkv_bias=True)
    model = transformers.ViTModel(cfg, add_pooling_layer=False).eval()
    names = list(model.state_dict().keys())
    # transformers renamed the encoder layers between releases: find the prefix and the projection names it uses
    pre = 'layers' if any(k.startswith('layers.0.') for k in names) else 'encoder.layer'
    new_style = any('.attention.q_proj.' in k for k in names)
    q, k, v, o = (('attention.q_proj', 'attention.k_proj', 'attention.v_proj', 'attention.o_proj') if new_style else
                  ('attention.attention.query', 'attention.attention.key', 'attention.attention.value', 'attention.output.dense'))
    fc1, fc2 = ('mlp.fc1', 'mlp.fc2') if new_style else ('intermediate.dense', 'output.dense')
    hf = {'embeddings.cls_token': sd['cls_token'], 'embeddings.position_embeddings': sd['pos_embed'],
          'embeddings.patch_embeddings.projection.weight': sd['patch_embed.proj.weight'],
          'embeddings.patch_embeddings.projection.bias': sd['patch_embed.proj.bias'],
          'layernorm.weight': sd['norm.weight'], 'layernorm.bias': sd['norm.bias']}
    for i in range(depth):
        w, b = sd[f'blocks.{i}.attn.qkv.weight'], sd[f'blocks.{i}.attn.qkv.bias']
        for j, name in enumerate((q, k, v)):            # the three thirds of Attention.qkv, in order
            hf[f'{pre}.{i}.{name}.weight'] = w[j * dim:(j + 1) * dim]
            hf[f'{pre}.{i}.{name}.bias'] = b[j * dim:(j + 1) * dim]
        for theirs, ours in ((o, 'attn.proj'), ('layernorm_before', 'norm1'), ('layernorm_after', 'norm2'),
                             (fc1, 'mlp.fc1'), (fc2, 'mlp.fc2')):
            for p in ('weight', 'bias'):
                hf[f'{pre}.{i}.{theirs}.{p}'] = sd[f'blocks.{i}.{ours}.{p}']
    model.load_state_dict(hf, strict=True)
    layers = model.layers if hasattr(model, 'layers') else model.encoder.layer
    key_proj = layers[-1].attention.k_proj if new_style else layers[-1].attention.attention.key
    return model, layers[-1].layernorm_before, key_proj


@pytest.mark.parametrize('arch', ['vits8', (128, 3, 2, 8)])
def test_oracle_vit_matches_transformers_vit(arch):
    dim, depth, heads, patch = vt.ARCHS[arch] if isinstance(arch, str) else arch
    sd = vt.synthetic_state_dict(arch, 3)
    g = torch.Generator().manual_seed(7)
    for k in sd:                        # the synthetic recipe has zero biases / unit LayerNorm gains: make every term count
        if k.endswith('.bias'):
            sd[k] = 0.1 * torch.randn(sd[k].shape, generator=g)
        elif 'norm' in k and k.endswith('.weight'):
            sd[k] = 1.0 + 0.2 * torch.randn(sd[k].shape, generator=g)
    oracle = dino_vit.build_vit(arch, sd)
    try:
        hf, ln_before, key_proj = _hf_from_dino(sd, dim, depth, heads, patch)
    except (RuntimeError, KeyError, AttributeError) as e:     # an unexpected transformers layout: nothing to compare with
        pytest.skip(f'transformers ViT layout not recognised: {e}')
    x = torch.randn(2, 3, 224, 224, generator=g)
    with torch.no_grad():
        out = hf(pixel_values=x, output_hidden_states=True)
        cls_ref = oracle(x)
        stream_ref = oracle.tokens_before_block(x, depth - 1)
        k_ref = oracle.last_block_k(x)
        k_hf = key_proj(ln_before(out.hidden_states[depth - 1]))
    # fp32 on both sides, different operation order: agreement to a few ulps of the largest value
    assert float((out.hidden_states[depth - 1] - stream_ref).abs().max()) <= 2e-5 * float(stream_ref.abs().max())
    assert float((k_hf - k_ref).abs().max()) <= 2e-5 * float(k_ref.abs().max())
    assert float((out.last_hidden_state[:, 0] - cls_ref).abs().max()) <= 2e-5 * float(cls_ref.abs().max())
