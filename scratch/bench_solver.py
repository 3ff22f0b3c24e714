"""HBM roofline of the per-SDE-step solver kernel (mgx_flow_step_fwd): algorithmic 12 B/element rollout (x fp32 in, v bf16 in,
noise bf16 in, x' fp32 out), 3,145,728 B per image-step."""
import sys, torch
sys.path.insert(0, ".")
from mixgrpo_amd import sampling_utils as SU
dev = "cuda"
sig = SU.sd3_time_shift(3.0, torch.linspace(1, 0, 26))
for B in (8, 64, 512):
    x = torch.randn(B, 4096, 64, device=dev); v = torch.randn(B, 4096, 64, device=dev).bfloat16()
    nz = torch.randn(B, 4096, 64, device=dev).bfloat16()
    for _ in range(3): SU.flow_grpo_step(v, x, 0.7, sig, 3, None, noise=nz)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    n = 20
    e0.record()
    for _ in range(n): SU.flow_grpo_step(v, x, 0.7, sig, 3, None, noise=nz)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / n
    by = B * 4096 * 64 * 12
    print(f"flow_grpo_step B={B}: {ms*1e3:.1f} us per call (host wrapper included), {by/ms/1e6:.0f} GB/s algorithmic (12 B/elem)")
