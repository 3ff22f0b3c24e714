"""A/B on one box: mgx_attn_fwd (multiply-add softmax) against mgx_attn_fwd_log2 (accumulator-initialised softmax, attn_fwd64q)
at the FLUX shapes of the train step (H 24, S 4608; B = 4 micro-batch, B = 12 rollout batch).  Arms alternate."""
import json, math, sys, torch
sys.path.insert(0, ".")
from mixgrpo_amd import ops
C = 1.4426950408889634 / math.sqrt(128)
H, S = 24, 4608


def t(fn, n):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


for B in (4, 12):
    g = torch.Generator(device="cuda").manual_seed(B)
    q, k, v = (torch.randn(B, H, S, 128, device="cuda", generator=g).bfloat16() for _ in range(3))
    q2 = (q.float() * C).bfloat16()
    vt = v.transpose(-1, -2).contiguous()
    O = torch.empty(B, S, H * 128, device="cuda", dtype=torch.bfloat16)
    lse = torch.empty(B, H, S, device="cuda")
    plain = lambda: ops.attn_fwd(q, k, vt, O, lse, B, H, S, S, H * 128, S * H * 128, 1 / math.sqrt(128))
    acc = lambda: ops.attn_fwd_log2(q2, k, vt, O, lse, B, H, S, S, H * 128, S * H * 128)
    r = [[], []]
    for rep in range(4):
        r[0].append(t(plain, 10)); r[1].append(t(acc, 10))
    a, b = min(r[0]), min(r[1])
    fl = 4.0 * B * H * S * S * 128 / 1e9
    print(json.dumps(dict(B=B, ms_plain=round(a, 4), ms_log2=round(b, 4), tf_plain=round(fl / a), tf_log2=round(fl / b),
                          gain=round(a / b, 4))), flush=True)
