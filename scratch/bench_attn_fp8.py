"""bf16 vs fp8 attention forward at the FLUX shape (B=8, H=24, S=4608): kernel time, quantise time."""
import sys, math, torch
sys.path.insert(0, ".")
from mixgrpo_amd import ops
torch.manual_seed(0)
B, H, S = 8, 24, 4608
Sp = S
dev = "cuda"
q = torch.randn(B, H, S, 128, device=dev).bfloat16(); k = torch.randn(B, H, S, 128, device=dev).bfloat16()
v = torch.randn(B, H, S, 128, device=dev).bfloat16()
vt = v.transpose(-1, -2).contiguous()
O = torch.empty(B, S, H * 128, device=dev, dtype=torch.bfloat16); lse = torch.empty(B, H, S, device=dev)
O8 = torch.empty_like(O); lse8 = torch.empty_like(lse)
u8 = lambda *s: torch.empty(*s, dtype=torch.uint8, device=dev)
Q8, K8, V8t = u8(B, H, S, 128), u8(B, H, S, 128), u8(B, H, 128, Sp)
amax = torch.empty(3 * B * H, device=dev)
def t(fn, n=5):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
sc = 1 / math.sqrt(128)
fl = 4.0 * B * H * S * S * 128
ms = t(lambda: ops.attn_fwd(q, k, vt, O, lse, B, H, S, Sp, H * 128, S * H * 128, sc))
print(f"bf16 attn fwd : {ms:.3f} ms  {fl/ms/1e9:.0f} TFLOP/s")
mq = t(lambda: ops.attn_fp8_quantize(q, k, vt, Q8, K8, V8t, amax, B, H, S, Sp))
by = 3 * B * H * S * 128 * (2 + 2 + 1)
print(f"fp8 quantise  : {mq:.3f} ms  {by/mq/1e6:.0f} GB/s (amax pass + quantise pass)")
m8 = t(lambda: ops.attn_fwd_fp8(Q8, K8, V8t, amax, O8, lse8, B, H, S, Sp, H * 128, S * H * 128, sc))
print(f"fp8 attn fwd  : {m8:.3f} ms  {fl/m8/1e9:.0f} TFLOP/s; with quantise {fl/(m8+mq)/1e9:.0f} TFLOP/s")
print("rel diff fp8 vs bf16:", ((O8.float() - O.float()).norm() / O.float().norm()).item(),
      "lse max diff", (lse8 - lse).abs().max().item())
