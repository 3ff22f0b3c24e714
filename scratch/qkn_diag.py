"""Price list of the EPI_QKNORM epilogue (mgx_linear_qk_norm_rope) at the rollout's shape (36864 tokens, H 24, K 3072): the plain
q | k projection (mgx_gemm_bf16, N = 6144) + norm pass against the fused launch, and a build under -fno-slp-vectorize.
(The timing-only variants of profiles/r04_qknorm_epilogue_prices.log -- no cos / sin loads, no exchange barrier -- were
`#ifdef MGX_DIAG_QKN_NOLOAD / _NOBAR` blocks in qknorm_epilogue; they are out of the product source again: to repeat them, stub
`load_cs` with constants / drop the s_barrier in a scratch copy.)  Builds its own libraries into scratch/."""
import ctypes as C, json, os, subprocess, sys, torch
sys.path.insert(0, ".")
from mixgrpo_amd import _lib, ops
from mixgrpo_amd.ops import Rows
CS = "mixgrpo_amd/csrc"
def build(tag, flags):
    so = f"scratch/libqkn_{tag}.so"
    subprocess.check_call(["hipcc", "-O3", "-std=c++17", "-fPIC", "-shared", "--offload-arch=gfx950"] + flags +
                          [f"{CS}/gemm.hip", f"{CS}/api.hip", "-o", so])
    lib = C.CDLL(so)
    res, args = _lib.SIGNATURES["mgx_linear_qk_norm_rope"]
    lib.mgx_linear_qk_norm_rope.restype, lib.mgx_linear_qk_norm_rope.argtypes = res, args
    return lib
variants = {}
noslp = build("noslp", ["-fno-slp-vectorize"])
torch.manual_seed(0)
B, rows, H, K, S = 8, 4608, 24, 3072, 4608
d, tokens = H * 128, B * rows
X = (torch.randn(tokens, K, device="cuda") * 0.7).bfloat16()
W = (torch.randn(2 * d, K, device="cuda") * 0.02).bfloat16()
bias = torch.zeros(2 * d, device="cuda", dtype=torch.bfloat16)
wq = torch.ones(128, device="cuda"); wk = torch.ones(128, device="cuda")
cos = torch.rand(S, 64, device="cuda").repeat_interleave(2, dim=1).contiguous(); sin = torch.rand(S, 64, device="cuda").repeat_interleave(2, dim=1).contiguous()
pairs = ops.rope_pair_table(cos, sin)

Q = torch.empty(B, H, S, 128, device="cuda", dtype=torch.bfloat16); Kt = torch.empty_like(Q)
qkv = torch.empty(tokens, 3 * d, device="cuda", dtype=torch.bfloat16)
st = torch.cuda.current_stream().cuda_stream
p = lambda t: t.data_ptr()
def t(fn, n=10):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
ops.GEMM_STREAM_K = False
cases = {"plain_gemm_N6144": lambda: ops.gemm(Rows.of(X), W, bias, Rows(qkv, tokens, 3 * d), 2 * d, K),
         "norm_pass_qk_only": lambda: ops.qk_norm_rope(qkv, wq, wk, cos, sin, Q, Kt, None, B, H, S, S, rows, 0),
         "fused_general_tables": lambda: ops.linear_qk_norm_rope(X, W, bias, wq, wk, cos, sin, Q, Kt, B, H, S, rows, 0, K),
         "fused_pair_table": lambda: ops.linear_qk_norm_rope(X, W, bias, wq, wk, cos, sin, Q, Kt, B, H, S, rows, 0, K, pairs=pairs)}
for name, lib in variants.items():
    cases["fused_" + name] = (lambda lib=lib: lib.mgx_linear_qk_norm_rope(p(X), p(W), p(bias), p(wq), p(wk), p(cos), p(sin), None, p(Q), p(Kt),
                                                                          B, H, S, rows, 0, K, K, K, 1.0, st))
cases["fused_general_noslp"] = lambda: noslp.mgx_linear_qk_norm_rope(p(X), p(W), p(bias), p(wq), p(wk), p(cos), p(sin), None, p(Q), p(Kt), B, H, S, rows, 0, K, K, K, 1.0, st)
res = {k: [] for k in cases}
for rep in range(3):
    for k, fn in cases.items():
        res[k].append(t(fn))
for k, v in res.items():
    print(json.dumps({"case": k, "ms": round(min(v), 4)}), flush=True)
