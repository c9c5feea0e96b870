"""EXPERIMENT: does a per-XCD start offset (MX_V3_STAGGER_US, read once at first launch) shorten the persistent 256x256
GEMM by spreading the epilogue's write bursts?  Runs each setting in a child process."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
CODE = r'''
import sys, torch
sys.path.insert(0, %r)
from sduss_amd import ops
dev = "cuda:0"
g = torch.Generator(device=dev).manual_seed(0)
def bench(fn, iters=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / iters * 1e3
out = []
for kind, m, n, k in (("gemm", 32768, 4608, 1536), ("geglu", 8192, 10240, 1280), ("gemm", 8192, 3840, 1280), ("gemm", 32768, 1536, 6144)):
    a = torch.randn(m, k, device=dev, generator=g).to(torch.bfloat16)
    w = (torch.randn(n, k, device=dev, generator=g) * k ** -0.5).to(torch.bfloat16)
    bias = torch.randn(n, device=dev, generator=g)
    t = bench((lambda: ops.gemm(a, w, bias, geglu=True)) if kind == "geglu" else (lambda: ops.gemm(a, w, bias)))
    out.append("%%s M%%d N%%d K%%d %%.1f us %%.0f TF" %% (kind, m, n, k, t, 2.0 * m * n * k / t / 1e6))
print(" | ".join(out))
''' % ROOT
for us in (0, 10, 20, 30, 45, 60):
    env = dict(os.environ, MX_V3_STAGGER_US=str(us))
    r = subprocess.run([sys.executable, "-c", CODE], env=env, capture_output=True, text=True)
    print(f"stagger {us:3d} us: {r.stdout.strip()} {r.stderr.strip()[-200:] if r.returncode else ''}", flush=True)
