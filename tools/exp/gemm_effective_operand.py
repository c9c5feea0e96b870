import ctypes as C, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from sduss_amd import lib
l = lib.load()
def rt(t): return t.to(torch.bfloat16).to(torch.float32)
m, n, k1, k2 = 256, 640, 256, 256
g = torch.Generator().manual_seed(1)
a1 = rt(torch.randn(m, k1, generator=g)); a2 = rt(torch.randn(m, k2, generator=g))
w = rt(torch.randn(n, k1 + k2, generator=g) * (k1 + k2) ** -0.5)
a1g, a2g, wg = a1.bfloat16().cuda(), a2.bfloat16().cuda(), w.bfloat16().cuda()
out = torch.empty(m, n, dtype=torch.bfloat16, device="cuda")
d = lib.GemmDesc()
d.a, d.w, d.c = a1g.data_ptr(), wg.data_ptr(), out.data_ptr()
d.M, d.N, d.K, d.lda, d.ldc = m, n, k1 + k2, k1, n
d.a2, d.lda2, d.k_split = a2g.data_ptr(), k2, k1
lib.check(l.mx_gemm(lib.current_stream(), C.byref(d)), "mx_gemm split A")
torch.cuda.synchronize()
o = out.float().cpu()
A = torch.cat([a1, a2], dim=1)
print("max err", (o - A @ w.t()).abs().max().item())
X = o @ torch.linalg.pinv(w.t())        # effective A operand [m, K]
for t in range((k1 + k2) // 64):
    xs = X[:, t * 64:(t + 1) * 64]
    res = []
    for u in range((k1 + k2) // 64):
        c = torch.nn.functional.cosine_similarity(xs.flatten(), A[:, u * 64:(u + 1) * 64].flatten(), dim=0).item()
        res.append(round(c, 2))
    print(f"K tile {t}: cos with A tiles {res}; norm ratio {xs.norm().item() / A[:, t * 64:(t + 1) * 64].norm().item():.2f}")
