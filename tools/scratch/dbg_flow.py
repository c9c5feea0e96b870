import torch, sys
sys.path.insert(0, '/root/repo')
from oracle import scheduler_ref
from sduss_amd import ops
g = torch.Generator().manual_seed(4)
n = 3
for dtype in (torch.float32, torch.bfloat16):
    lat = torch.randn(n, 16, 8, 8, generator=g).to(dtype)
    noise = torch.randn(2 * n, 16, 8, 8, generator=g).to(dtype)
    sig = torch.tensor([1.0, 0.7, 0.2]); sig_next = torch.tensor([0.95, 0.6, 0.0])
    comb = scheduler_ref.cfg_combine(noise, 7.0)
    want = scheduler_ref.flow_match_step(comb, lat, sig, sig_next)
    got = ops.cfg_flow_step_(noise.cuda(), lat.cuda().clone(), sig, sig_next, 7.0).cpu()
    bad = (got.float() != want.float())
    print(dtype, "mismatch", bad.sum().item(), "of", bad.numel())
    if bad.any():
        i = bad.flatten().nonzero()[0].item()
        u, t = noise[:n].flatten()[i].item(), noise[n:].flatten()[i].item()
        print(" u", u, "t", t, "comb", comb.flatten()[i].item(), "lat", lat.flatten()[i].item(), "got", got.flatten()[i].item(), "want", want.flatten()[i].item())
        # variants
        import numpy as np
        f = np.float32
        d = f(t) - f(u); gd = f(7.0) * d; c1 = f(u) + gd
        print(" np separate:", c1, " fma-ish:", f(np.float64(u) + np.float64(7.0)*np.float64(d)))
    # g=0 path: only the step
    got2 = ops.cfg_flow_step_(comb.cuda(), lat.cuda().clone(), sig, sig_next, 0.0).cpu()
    print("  step-only mismatch", (got2.float() != want.float()).sum().item())
