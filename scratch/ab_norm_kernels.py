"""Same-process A/B of `mgx_qk_norm_rope_fwd` from round 3's library (scratch/libmixgrpo_old.so: one token per pass) and the in-tree
one (four tokens' loads in flight per pass); outputs must be BIT-IDENTICAL (same arithmetic, same reduction order)."""
import ctypes as C, os, sys, torch
sys.path.insert(0, ".")
from mixgrpo_amd import _lib
new = _lib.lib()
old = C.CDLL(os.path.join("scratch", "libmixgrpo_old.so"))
res, args = _lib.SIGNATURES["mgx_qk_norm_rope_fwd"]
old.mgx_qk_norm_rope_fwd.restype = res
old.mgx_qk_norm_rope_fwd.argtypes = args
torch.manual_seed(0)
dev = "cuda"
st = torch.cuda.current_stream().cuda_stream
H, hd = 24, 128


def t(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


for (B, S, rows, s0, emit) in [(8, 4608, 4608, 0, False), (8, 4608, 4096, 512, False), (8, 4608, 512, 0, False), (4, 4608, 4608, 0, True),
                               (4, 4608, 4096, 512, True), (1, 4608, 4608, 0, False)]:
    d = H * hd
    qkv = torch.randn(B * rows, 3 * d, device=dev).bfloat16()
    wq = 1 + 0.1 * torch.randn(hd, device=dev)
    wk = 1 + 0.1 * torch.randn(hd, device=dev)
    cos, sin = torch.randn(S, hd, device=dev), torch.randn(S, hd, device=dev)
    outs = []
    for lib in (old, new):
        Q = torch.zeros(B, H, S, hd, device=dev, dtype=torch.bfloat16)
        K = torch.zeros_like(Q)
        Vt = torch.zeros(B, H, hd, S, device=dev, dtype=torch.bfloat16)
        V = torch.zeros_like(Q) if emit else None
        Qt = torch.zeros_like(Vt) if emit else None
        Kt = torch.zeros_like(Vt) if emit else None
        p = lambda x: None if x is None else x.data_ptr()
        fn = lambda lib=lib, Q=Q, K=K, Vt=Vt, V=V, Qt=Qt, Kt=Kt: lib.mgx_qk_norm_rope_fwd(
            p(qkv), 3 * d, p(wq), p(wk), p(cos), p(sin), p(Q), p(K), p(Vt), p(V), p(Qt), p(Kt), B, H, S, S, rows, s0, st)
        assert fn() == 0
        torch.cuda.synchronize()
        outs.append(((Q, K, Vt, V, Qt, Kt), fn))
    same = all((a is None and b is None) or torch.equal(a, b) for a, b in zip(outs[0][0], outs[1][0]))
    r = [[], []]
    for rep in range(3):
        r[0].append(t(outs[0][1]))
        r[1].append(t(outs[1][1]))
    a, b_ = min(r[0]), min(r[1])
    nbytes = B * rows * d * 2 * (3 + 3 + (3 if emit else 0))
    print(f"B{B} rows{rows} s0 {s0} emit_t {emit}: identical={same}  r03 {a * 1e3:.1f} us {nbytes / a / 1e9:.2f} TB/s | r04 {b_ * 1e3:.1f} us "
          f"{nbytes / b_ / 1e9:.2f} TB/s ({100 * (a / b_ - 1):+.1f} %)", flush=True)
    assert same

# ---- mgx_ln_modulate_bwd: next row prefetched (packed) while the current row's three reductions run; bit-identical
res, args = _lib.SIGNATURES["mgx_ln_modulate_bwd"]
old.mgx_ln_modulate_bwd.restype = res
old.mgx_ln_modulate_bwd.argtypes = args
old.mgx_ln_modulate_bwd_workspace.restype = C.c_long
old.mgx_ln_modulate_bwd_workspace.argtypes = [C.c_long, C.c_long, C.c_int]
D = 3072
for (B, rows, S, off, acc) in [(4, 4608, 4608, 0, 1), (4, 4096, 4608, 512, 1), (4, 512, 4608, 0, 1), (4, 4096, 4608, 512, 0)]:
    M = B * rows
    dy = torch.randn(M, D, device=dev).bfloat16()
    xfull = torch.randn(B, S, D, device=dev).bfloat16()
    x = xfull[0, off:]
    scale = (0.2 * torch.randn(B, 6 * D, device=dev)).bfloat16()
    dx0 = torch.randn(B, S, D, device=dev).bfloat16()
    ws = torch.empty(new.mgx_ln_modulate_bwd_workspace(M, rows, D), dtype=torch.float32, device=dev)
    outs = []
    for lib in (old, new):
        dxf = dx0.clone()
        dsh = torch.zeros(B, 6 * D, device=dev, dtype=torch.bfloat16)
        fn = lambda lib=lib, dxf=dxf, dsh=dsh: lib.mgx_ln_modulate_bwd(
            dy.data_ptr(), D, x.data_ptr(), D, rows, S * D, scale[:, D:].data_ptr(), 6 * D, dxf[0, off:].data_ptr(), D, rows, S * D, acc,
            dsh[:, 0:].data_ptr(), dsh[:, D:].data_ptr(), ws.data_ptr(), M, D, st)
        assert fn() == 0
        torch.cuda.synchronize()
        outs.append((dxf.clone(), dsh.clone(), fn))
    same = torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])
    r = [[], []]
    for rep in range(3):
        r[0].append(t(outs[0][2]))
        r[1].append(t(outs[1][2]))
    a, b_ = min(r[0]), min(r[1])
    nbytes = M * D * 2 * (3 + acc)
    print(f"ln_mod_bwd B{B} rows{rows} acc{acc}: identical(first call)={same}  r03 {a * 1e3:.1f} us {nbytes / a / 1e9:.2f} TB/s | r04 {b_ * 1e3:.1f} us "
          f"{nbytes / b_ / 1e9:.2f} TB/s ({100 * (a / b_ - 1):+.1f} %)", flush=True)
    assert same

# ---- mgx_qk_norm_rope_bwd: four tokens' loads in flight per pass; bit-identical
res, args = _lib.SIGNATURES["mgx_qk_norm_rope_bwd"]
old.mgx_qk_norm_rope_bwd.restype = res
old.mgx_qk_norm_rope_bwd.argtypes = args
for (B, S, rows, s0) in [(4, 4608, 4608, 0), (4, 4608, 4096, 512), (4, 4608, 512, 0)]:
    d = H * hd
    qkv = torch.randn(B * rows, 3 * d, device=dev).bfloat16()
    wq = 1 + 0.1 * torch.randn(hd, device=dev)
    wk = 1 + 0.1 * torch.randn(hd, device=dev)
    cos, sin = torch.randn(S, hd, device=dev), torch.randn(S, hd, device=dev)
    dQ, dK, dV = (torch.randn(B, H, S, hd, device=dev).bfloat16() for _ in range(3))
    ws = torch.empty(new.mgx_qk_norm_rope_bwd_workspace(B, H, rows), dtype=torch.float32, device=dev)
    outs = []
    for lib in (old, new):
        dqkv = torch.zeros(B * rows, 3 * d, device=dev, dtype=torch.bfloat16)
        gq, gk = torch.zeros(hd, device=dev), torch.zeros(hd, device=dev)
        fn = lambda lib=lib, dqkv=dqkv, gq=gq, gk=gk: lib.mgx_qk_norm_rope_bwd(
            qkv.data_ptr(), 3 * d, wq.data_ptr(), wk.data_ptr(), cos.data_ptr(), sin.data_ptr(), dQ.data_ptr(), dK.data_ptr(), dV.data_ptr(),
            dqkv.data_ptr(), 3 * d, gq.data_ptr(), gk.data_ptr(), ws.data_ptr(), B, H, S, S, rows, s0, st)
        assert fn() == 0
        torch.cuda.synchronize()
        outs.append((dqkv.clone(), gq.clone(), gk.clone(), fn))
    same = all(torch.equal(outs[0][k], outs[1][k]) for k in range(3))
    r = [[], []]
    for rep in range(3):
        r[0].append(t(outs[0][3]))
        r[1].append(t(outs[1][3]))
    a, b_ = min(r[0]), min(r[1])
    nbytes = B * rows * d * 2 * (2 + 3 + 3)
    print(f"qk_norm_rope_bwd B{B} rows{rows}: identical(first call)={same}  r03 {a * 1e3:.1f} us {nbytes / a / 1e9:.2f} TB/s | r04 {b_ * 1e3:.1f} us "
          f"{nbytes / b_ / 1e9:.2f} TB/s ({100 * (a / b_ - 1):+.1f} %)", flush=True)
    assert same
