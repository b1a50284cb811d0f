#!/usr/bin/env python3
"""What unmodified PyTorch-ROCm does on the same MI355X for the same work: a plain torch ViT-S/8 (or ViT-B/8) written
the way the DINO model the reference fetches is written -- nn.Conv2d patch embedding, nn.LayerNorm, nn.Linear, explicit
softmax(q k^T / 8) v attention (`--sdpa`: F.scaled_dot_product_attention instead), exact GELU, all 12 blocks, forward hook
on the last qkv -- run like the reference runs it on a GPU (infer.py:173-177, 309): fp16 autocast, no_grad, mini-batches of
512 x 512 slices (N = 4097), hooked output copied to the host as fp16.  Random weights, random slices; informative only
(slices/s next to bench.py's number; not part of the bench contract, no parity claim, nothing from oracle/).

    python tools/stock_pipeline.py [--arch vits8|vitb8] [--batch 8] [--slices 64] [--sdpa] [--keep-on-device]
"""
import argparse
import time

import torch
import torch.nn as nn
import torch.nn.functional as F


class Attention(nn.Module):
    def __init__(self, dim, heads, sdpa):
        super().__init__()
        self.num_heads, self.scale, self.sdpa = heads, (dim // heads) ** -0.5, sdpa
        self.qkv = nn.Linear(dim, 3 * dim)
        self.proj = nn.Linear(dim, dim)

    def forward(self, x):
        b, n, c = x.shape
        qkv = self.qkv(x).reshape(b, n, 3, self.num_heads, c // self.num_heads).permute(2, 0, 3, 1, 4)
        q, k, v = qkv[0], qkv[1], qkv[2]
        if self.sdpa:
            o = F.scaled_dot_product_attention(q, k, v)
        else:
            attn = (q @ k.transpose(-2, -1)) * self.scale
            o = attn.softmax(dim=-1) @ v
        return self.proj(o.transpose(1, 2).reshape(b, n, c))


class Block(nn.Module):
    def __init__(self, dim, heads, sdpa):
        super().__init__()
        self.norm1, self.norm2 = nn.LayerNorm(dim, eps=1e-6), nn.LayerNorm(dim, eps=1e-6)
        self.attn = Attention(dim, heads, sdpa)
        self.fc1, self.fc2 = nn.Linear(dim, 4 * dim), nn.Linear(4 * dim, dim)

    def forward(self, x):
        x = x + self.attn(self.norm1(x))
        return x + self.fc2(F.gelu(self.fc1(self.norm2(x))))


class ViT(nn.Module):
    def __init__(self, dim, depth, heads, patch, tokens, sdpa):
        super().__init__()
        self.embed = nn.Conv2d(3, dim, patch, patch)
        self.cls = nn.Parameter(torch.zeros(1, 1, dim))
        self.pos = nn.Parameter(torch.randn(1, tokens, dim) * 0.02)
        self.blocks = nn.ModuleList([Block(dim, heads, sdpa) for _ in range(depth)])
        self.norm = nn.LayerNorm(dim, eps=1e-6)

    def forward(self, img):
        x = self.embed(img).flatten(2).transpose(1, 2)
        x = torch.cat((self.cls.expand(x.shape[0], -1, -1), x), dim=1) + self.pos
        for blk in self.blocks:
            x = blk(x)
        return self.norm(x)[:, 0]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--arch', default='vits8', choices=['vits8', 'vitb8'])
    ap.add_argument('--batch', type=int, default=8)
    ap.add_argument('--slices', type=int, default=64)
    ap.add_argument('--sdpa', action='store_true')
    ap.add_argument('--keep-on-device', action='store_true', help='skip the per-batch copy of the hooked tensor to the host')
    args = ap.parse_args()
    dim, depth, heads = (384, 12, 6) if args.arch == 'vits8' else (768, 12, 12)
    dev = torch.device('cuda', 0)
    torch.manual_seed(0)
    model = ViT(dim, depth, heads, 8, 4097, args.sdpa).to(dev).eval()
    hooked = []
    model.blocks[-1].attn.qkv.register_forward_hook(
        lambda m, i, o: hooked.append(o.half() if args.keep_on_device else o.cpu().half()))
    imgs = torch.randn(args.slices, 1, 512, 512).expand(-1, 3, -1, -1)          # host tensor, like the reference's

    def run():
        hooked.clear()
        with torch.no_grad(), torch.autocast('cuda', dtype=torch.float16):
            for b0 in range(0, args.slices, args.batch):
                model(imgs[b0:b0 + args.batch].to(dev))
        torch.cuda.synchronize()

    run()                                                                        # warm-up (library autotuning, allocator)
    t0 = time.perf_counter()
    run()
    dt = time.perf_counter() - t0
    print(f'stock PyTorch-ROCm {args.arch} N=4097, fp16 autocast, batch {args.batch}, '
          f'{"SDPA" if args.sdpa else "explicit softmax(q k^T) v"}, hooked qkv '
          f'{"kept on the device" if args.keep_on_device else "copied to the host"}: '
          f'{args.slices / dt:.1f} slices/s ({dt / args.slices * 1e3:.2f} ms per slice)', flush=True)


if __name__ == '__main__':
    main()
