import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session", autouse=True)
def oracle_host_threads():
    """The fp32 torch oracles run fastest on ~32 host threads on the GPU box (256 CPUs visible, torch defaults to 128: an SDXL-base 1024^2 sample-forward takes
    57 s at 128 threads, 11.5 s at 32; SD3.5-medium 77 s vs 39.6 s: tools/exp/oracle_threads.py, profiles/r04_oracle_threads.txt).  Fewer where the host has fewer."""
    import torch
    n = torch.get_num_threads()
    torch.set_num_threads(min(n, 32))
    yield
    torch.set_num_threads(n)


@pytest.fixture(scope="session")
def cuda_device():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    return torch.device("cuda:0")


@pytest.fixture(scope="session")
def full_width_sdxl(cuda_device):
    """ONE set of full-width SDXL-base parameters (2.57 B, random init), their device-held form for the oracle (weights.params_as_held) and ONE
    packed model on the GPU, shared by every full-width forward of the session (generating, folding and packing them takes half a minute)."""
    import torch
    from oracle import sdxl_unet_ref as ref
    from sduss_amd.config import UNetConfig
    from sduss_amd.unet import MxUNet
    from sduss_amd.weights import params_as_held
    ocfg = ref.UNetConfig.sdxl_base()
    P = ref.fast_params(ocfg)
    held = params_as_held(UNetConfig.sdxl_base(), P)
    net = MxUNet(UNetConfig.sdxl_base(), P, device="cuda:0")
    yield ocfg, P, held, net
    del net
    torch.cuda.empty_cache()
