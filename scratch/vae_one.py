"""Scratch: a loop of FLUX-VAE decodes of one 128 x 128 latent for rocprofv3 --kernel-trace --stats."""
import sys, time
sys.path.insert(0, ".")
import torch
from mixgrpo_amd.vae import AutoencoderKL
m = AutoencoderKL(device="cuda").init_synthetic(seed=1)
z = torch.randn(1, 16, 128, 128, device="cuda")
for _ in range(2):
    m.decode(z, return_dict=False)
torch.cuda.synchronize()
t0 = time.perf_counter()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 5
for _ in range(n):
    m.decode(z, return_dict=False)
torch.cuda.synchronize()
print(f"decode {(time.perf_counter() - t0) / n * 1e3:.2f} ms per 1024^2 image", flush=True)
