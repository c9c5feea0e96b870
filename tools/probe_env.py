#!/usr/bin/env python3
"""One-shot probe of the GPU box: is the third-party `diffusers` package
importable, and are the public HF snapshots of the two models on disk?  If both were present, the stock diffusers CPU
pipeline would be the end-to-end parity oracle and the timed CPU baseline; otherwise the oracle stays this repo's torch
restatement.  Prints one JSON object; writes it to gpurun_out/probe_env.json when that directory exists."""
import glob
import importlib
import json
import os
import platform

out = {"python": platform.python_version(), "cpu_count": os.cpu_count()}
try:
    with open("/proc/cpuinfo") as f:
        out["cpu_model"] = next((l.split(":", 1)[1].strip() for l in f if l.startswith("model name")), None)
    with open("/proc/meminfo") as f:
        out["mem_total_gib"] = round(int(f.readline().split()[1]) / 2 ** 20, 1)
except OSError:
    pass
for mod in ("diffusers", "xformers", "transformers", "accelerate", "safetensors"):
    try:
        m = importlib.import_module(mod)
        out[mod] = getattr(m, "__version__", "present")
    except Exception as e:  # noqa: BLE001
        out[mod] = f"absent ({type(e).__name__})"
roots = [os.environ.get("HF_HOME"), os.environ.get("HF_HUB_CACHE"), os.environ.get("TRANSFORMERS_CACHE"),
         os.path.expanduser("~/.cache/huggingface"), "/workspace/huggingface", "/root/.cache/huggingface", "/data", "/models", "/mnt"]
found = []
for r in roots:
    if r and os.path.isdir(r):
        for pat in ("**/models--stabilityai--stable-diffusion-xl-base-1.0", "**/models--stabilityai--stable-diffusion-3.5-medium",
                    "**/unet/diffusion_pytorch_model*.safetensors", "**/transformer/diffusion_pytorch_model*.safetensors"):
            found += glob.glob(os.path.join(r, pat), recursive=True)[:4]
out["hf_roots_present"] = [r for r in roots if r and os.path.isdir(r)]
out["model_snapshots_found"] = sorted(set(found))
try:
    import torch
    out["torch"] = torch.__version__
    out["torch_threads"] = torch.get_num_threads()
    if torch.cuda.is_available():
        p = torch.cuda.get_device_properties(0)
        out["gpu"] = {"name": p.name, "cus": p.multi_processor_count, "hbm_gib": round(p.total_memory / 2 ** 30, 1)}
except Exception as e:  # noqa: BLE001
    out["torch"] = f"error {e}"
s = json.dumps(out, indent=1)
print(s)
if os.path.isdir("gpurun_out"):
    with open("gpurun_out/probe_env.json", "w") as f:
        f.write(s)
