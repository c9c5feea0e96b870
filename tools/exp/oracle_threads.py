"""How many host threads the fp32 torch oracles want on the GPU box (round 4): one SD3.5-medium and one SDXL-base sample-forward at 1024^2 per thread count.
Usage on the GPU box: python tools/exp/oracle_threads.py"""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oracle import sd3_mmdit_ref, sdxl_unet_ref as ref

print("default threads", torch.get_num_threads(), "cpus", os.cpu_count(), flush=True)
ocfg3 = sd3_mmdit_ref.MMDiTConfig.sd35_medium(); P3 = sd3_mmdit_ref.init_params(ocfg3)
in3 = sd3_mmdit_ref.make_inputs(ocfg3, 1, 128, ctx_len=333)
ocfg = ref.UNetConfig.sdxl_base(); P = {k: v.float() for k, v in ref.fast_params(ocfg).items()}
inx = ref.make_inputs(ocfg, 1, 128)
for n in (128, 64, 32, 16):
    torch.set_num_threads(n)
    with torch.inference_mode():
        t0 = time.perf_counter(); sd3_mmdit_ref.mmdit_forward(P3, ocfg3, *in3); t1 = time.perf_counter()
        ref.unet_forward(P, ocfg, *inx); t2 = time.perf_counter()
    print(f"threads {n:4d}: SD3.5-medium {t1 - t0:6.1f} s | SDXL-base {t2 - t1:6.1f} s", flush=True)
