"""GEMM / conv microbenchmark over the SDXL step shapes (random bf16 data), through the C ABI.
Usage on the GPU box: python tools/gemm_bench.py   (SHAPES=small: the single-request shapes)"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from sduss_amd import ops  # noqa: E402

SHAPES = [  # (kind, M, N, K or (hw, cin))
    ("gemm", 8192, 1280, 1280), ("gemm", 8192, 1280, 5120), ("gemm", 8192, 3840, 1280), ("qkv", 8192, 3840, 1280), ("qkv", 32768, 1920, 640), ("geglu", 8192, 10240, 1280),
    ("gemm", 32768, 640, 640), ("geglu", 32768, 5120, 640), ("gemm", 32768, 640, 2560),
    ("conv", 8, 1280, (32, 1280)), ("conv", 8, 320, (128, 320)), ("conv", 8, 640, (64, 640)),
    # more launches of several rounds of 256 x 160 tiles (the 64 x 64 and 128 x 128 levels)
    ("gemm", 32768, 1920, 640), ("conv", 8, 320, (128, 640)), ("conv", 8, 640, (64, 1280)),
    # SD3.5-medium image-stream shapes (M = 8 x 4096 tokens, d = 1536)
    ("gemm", 32768, 4608, 1536), ("gemm", 32768, 1536, 1536), ("gemm", 32768, 6144, 1536), ("gemm", 32768, 1536, 6144),
]


def bench(fn, iters=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / iters * 1e-3


SMALL = [  # UNet batch 2 (one request): level-2 M = 2048, level-1 M = 8192
    ("gemm", 2048, 1280, 1280), ("gemm", 2048, 1280, 5120), ("gemm", 2048, 3840, 1280), ("geglu", 2048, 10240, 1280),
    ("gemm", 8192, 640, 640), ("geglu", 8192, 5120, 640), ("gemm", 8192, 640, 2560), ("gemm", 8192, 1920, 640),
    ("conv", 2, 1280, (32, 1280)), ("conv", 2, 320, (128, 320)), ("conv", 2, 640, (64, 640)),
]


def main():
    global SHAPES
    if os.environ.get("SHAPES") == "small":
        SHAPES = SMALL
    dev = "cuda:0"
    g = torch.Generator(device=dev).manual_seed(0)
    for kind, m, n, k in SHAPES:
        if kind == "conv":
            hw, cin = k
            x = torch.randn(m, hw, hw, cin, device=dev, generator=g).to(torch.bfloat16)
            w = (torch.randn(n, 9 * cin, device=dev, generator=g) * (9 * cin) ** -0.5).to(torch.bfloat16)
            bias = torch.randn(n, device=dev, generator=g)
            t = bench(lambda: ops.conv3x3(x, w, bias))
            fl = 2.0 * m * hw * hw * n * 9 * cin
            label = f"conv  B{m} {hw}x{hw} {cin}->{n}"
        else:
            a = torch.randn(m, k, device=dev, generator=g).to(torch.bfloat16)
            w = (torch.randn(n, k, device=dev, generator=g) * k ** -0.5).to(torch.bfloat16)
            bias = torch.randn(n, device=dev, generator=g)
            if kind == "qkv":                    # the fused q | k | v projection with its V^T epilogue (tokens per image: 1024 at 1280, 4096 at 640)
                rpb = 1024 if n == 3840 else 4096
                t = bench(lambda: ops.gemm_qkv(a, w, n // 3, 3, rpb, q_scale=ops.ATTN_QSCALE, bias=bias))
            elif kind == "geglu":
                t = bench(lambda: ops.gemm(a, w, bias, geglu=True))
            else:
                r = torch.randn(m, n, device=dev, generator=g).to(torch.bfloat16)
                t = bench(lambda: ops.gemm(a, w, bias, residual=r))
            fl = 2.0 * m * n * k
            label = f"{kind:5s} M{m} N{n} K{k}"
        print(f"{label:32s} {t * 1e6:9.1f} us  {fl / t / 1e12:8.1f} TFLOP/s")


if __name__ == "__main__":
    main()
