import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from sduss_amd.config import UNetConfig
from sduss_amd.unet import MxUNet
from sduss_amd.weights import synthetic_params
cfg = UNetConfig.sdxl_base()
P = synthetic_params(cfg, device="cuda:0")
net = MxUNet(cfg, P, device="cuda:0")
x = torch.randn(8,4,128,128,device="cuda").to(torch.bfloat16)
t = torch.full((8,), 500.0, device="cuda"); e = torch.randn(8,77,2048,device="cuda").to(torch.bfloat16); te=torch.randn(8,1280,device="cuda").to(torch.bfloat16); ti=torch.zeros(8,6,device="cuda")
y = net.forward_one(x,t,e,te,ti); torch.cuda.synchronize(); print("ok", float(y.float().abs().mean()))
