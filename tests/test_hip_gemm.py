"""GPU parity of `mgx_gemm_bf16` (csrc/gemm.hip) against a plain fp32 torch reference of the same op, through the
C ABI.  The shapes are chosen so that BOTH kernel families run: the persistent 256x256 kernel (>= 128 tiles) and the
128x128 kernel (small problems), with ragged M / N edges, row-batched A and C operands whose batch boundaries fall
inside tiles, and every fused epilogue.

Tolerance: bf16 operands, fp32 MFMA accumulation in a different summation order than the reference; the Linear's bf16
output y may differ from the rounded reference by one bf16 ulp on values that sit on a rounding boundary, and the fused
epilogue propagates such a flip with slope `amp` (GELU <= 1.13, gate: |gate|, GELU': <= 1.13):
|diff| <= 2^-7 * |ref| + 2 * amp * 2^-7 * |y| + 2e-3 everywhere (the flip, and one more rounding of the scaled term), and < 2 % of the elements differ at all.
fp32 accumulate (wgrad) outputs: relative L2 <= 1e-5."""
import pytest
import torch

pytestmark = pytest.mark.gpu

EPI_BIAS, EPI_GELU, EPI_GATE_RES, EPI_F32_ACC, EPI_DGELU = 0, 1, 2, 3, 4


def _gelu(x):
    return torch.nn.functional.gelu(x, approximate="tanh")


def _dgelu(x):
    x = x.double()
    k0, k1 = 0.7978845608028654, 0.044715
    u = k0 * (x + k1 * x ** 3)
    t = torch.tanh(u)
    return (0.5 * (1 + t) + 0.5 * x * (1 - t * t) * k0 * (1 + 3 * k1 * x * x)).float()


def _batched(M, cols, rpb, pad, dtype, gen, scale=1.0):
    """A row-batched matrix: batches of `rpb` rows, `pad` junk rows between batches.  Returns (storage, Rows, dense)."""
    from mixgrpo_amd.ops import Rows
    nb = (M + rpb - 1) // rpb
    store = (torch.randn(nb, rpb + pad, cols, generator=gen) * scale).to(dtype).cuda()
    dense = store[:, :rpb].reshape(nb * rpb, cols)[:M]
    return store, Rows(store, M, cols, rpb, (rpb + pad) * cols), dense


def _close_bf16(out, ref, y=None, amp=0.0):
    out, ref = out.float().cpu(), ref.float().cpu()
    assert torch.isfinite(out).all()
    diff = (out - ref).abs()
    tol = ref.abs() * 2.0 ** -7 + 2e-3
    if y is not None:
        tol = tol + 2 * torch.as_tensor(amp).float().cpu() * y.float().cpu().abs() * 2.0 ** -7
    assert (diff <= tol).all(), f"max excess {(diff - tol).max().item()}"
    assert (out != ref).float().mean().item() < 0.02


# (M, N, K): persistent kernel needs ceil(M/256)*ceil(N/256) >= 128 and (N % 256 == 0 or N >= 2048)
# (the 4th: 369 tiles on 256 workgroups and 7 K-tiles -- workgroups run on into a second tile with the LDS stage ring at
#  an odd position)
BIG = [(4096 + 37, 3072, 192), (6000, 2624, 128), (256 * 24, 2048, 64), (256 * 40 + 19, 2304, 448)]
SMALL = [(200, 136, 192), (1000, 64, 256), (513, 1032, 64)]


@pytest.mark.parametrize("M,N,K", BIG + SMALL)
@pytest.mark.parametrize("epi", [EPI_BIAS, EPI_GELU, EPI_GATE_RES, EPI_DGELU])
@pytest.mark.parametrize("batched", [0, 1000, 1024])   # rows per batch: 1024 keeps the 16-byte epilogue, 1000 does not
def test_gemm_epilogues(M, N, K, epi, batched):
    from mixgrpo_amd import ops
    from mixgrpo_amd.ops import Rows
    g = torch.Generator().manual_seed(M * 7 + N * 3 + K + epi)
    rpb = batched if batched else 1 << 40
    if batched:
        _, A_rows, A = _batched(M, K, rpb, 3, torch.bfloat16, g, 0.5)
        C_store, C_rows, C0 = _batched(M, N, rpb, 5, torch.bfloat16, g, 1.0)
    else:
        A = (torch.randn(M, K, generator=g) * 0.5).bfloat16().cuda()
        A_rows = Rows.of(A)
        C_store = torch.randn(M, N, generator=g).bfloat16().cuda()
        C_rows, C0 = Rows.of(C_store), C_store
    C0 = C0.clone()
    guard = C_store.clone()
    W = (torch.randn(N, K, generator=g) * 0.1).bfloat16().cuda()
    bias = (torch.randn(N, generator=g) * 0.2).bfloat16().cuda()
    nb = (M + rpb - 1) // rpb if batched else 1
    gate = (torch.randn(nb, N, generator=g)).bfloat16().cuda() if epi == EPI_GATE_RES else None
    aux = None
    if epi in (EPI_GELU, EPI_GATE_RES):
        aux = torch.full((M, N), 7.0, dtype=torch.bfloat16, device="cuda")
    elif epi == EPI_DGELU:
        aux = torch.randn(M, N, generator=g).bfloat16().cuda()
    ops.gemm(A_rows, W, bias, C_rows, N, K, epi, gate=gate, gate_ld=N, aux=aux)
    torch.cuda.synchronize()

    y = (A.float() @ W.float().t() + bias.float()).bfloat16().float()          # the Linear's bf16 output
    amp = 0.0
    if epi == EPI_BIAS:
        ref = y
    elif epi == EPI_GELU:
        ref, amp = _gelu(y), 1.2
    elif epi == EPI_GATE_RES:
        gsel = gate.float()[torch.arange(M, device="cuda") // rpb] if batched else gate.float()[0][None]
        ref, amp = C0.float() + (gsel * y).bfloat16().float(), gsel.abs() + 0.01
    else:
        ref, amp = y * _dgelu(aux.float().cpu()).cuda(), 1.2
    out = (C_store[:, :rpb].reshape(-1, N)[:M] if batched else C_store)
    _close_bf16(out, ref.bfloat16(), y, amp)
    if epi in (EPI_GELU, EPI_GATE_RES):
        _close_bf16(aux, y.bfloat16())                                           # saved pre-activation / pre-gate output
    if batched:                                                                  # padding rows between batches untouched
        assert torch.equal(C_store[:, rpb:], guard[:, rpb:])


@pytest.mark.parametrize("M,N,K,beta", [(3072, 4096 + 128, 320, 1.0), (2048 + 56, 3072, 128, 0.0), (300, 264, 192, 1.0)])
def test_gemm_f32_accumulate(M, N, K, beta):
    """wgrad form: C_f32 = beta * C + A @ W^T (no bias), both kernel families."""
    from mixgrpo_amd import ops
    from mixgrpo_amd.ops import Rows
    g = torch.Generator().manual_seed(11)
    A = (torch.randn(M, K, generator=g) * 0.5).bfloat16().cuda()
    W = (torch.randn(N, K, generator=g) * 0.5).bfloat16().cuda()
    C = torch.randn(M, N, generator=g).cuda()
    ref = beta * C.double() + A.double() @ W.double().t()
    ops.gemm(Rows.of(A), W, None, Rows.of(C), N, K, EPI_F32_ACC, beta=beta)
    torch.cuda.synchronize()
    err = ((C.double() - ref).norm() / ref.norm()).item()
    assert err < 1e-5, err


def test_gemm_full_size_linearity():
    """BASELINE-size check through a size-independent property: the GEMM is linear in A, so gemm(A1 + A2) with bf16-exact
    operands equals gemm(A1) + gemm(A2) up to the output rounding (M = 8*4608 joint rows, FLUX d = 3072)."""
    from mixgrpo_amd import ops
    from mixgrpo_amd.ops import Rows
    M, N, K = 8 * 4608, 3072, 3072
    g = torch.Generator(device="cuda").manual_seed(5)
    # small integers / 8: sums are exact in bf16, products exact in fp32
    A1 = (torch.randint(-4, 5, (M, K), generator=g, device="cuda").float() / 8).bfloat16()
    A2 = (torch.randint(-4, 5, (M, K), generator=g, device="cuda").float() / 8).bfloat16()
    W = (torch.randint(-2, 3, (N, K), generator=g, device="cuda").float() / 4).bfloat16()
    outs = []
    for A in (A1, A2, (A1.float() + A2.float()).bfloat16()):
        C = torch.empty(M, N, dtype=torch.float32, device="cuda")
        ops.gemm(Rows.of(A), W, None, Rows.of(C), N, K, EPI_F32_ACC, beta=0.0)
        outs.append(C)
    torch.cuda.synchronize()
    assert torch.equal(outs[0] + outs[1], outs[2])          # every partial sum is exactly representable: bit-exact
    for rows in (slice(0, 512), slice(M // 2 - 256, M // 2 + 256), slice(M - 512, M)):     # first, middle and last tiles
        assert torch.equal(outs[0][rows], A1[rows].float() @ W.float().t())                 # of the workgroups' lists


def test_small_tile_kernel_stays_selectable():
    """`MGX_GEMM_MODE=0` (read once per process; debugging) forces the 128x128 kernel everywhere: it must keep producing the
    same Linear as the default persistent ping-pong kernel (both within the bf16 tolerance of the fp32 reference)."""
    import os, subprocess, sys
    code = (
        "import sys, torch; sys.path.insert(0, '.')\n"
        "from mixgrpo_amd import ops; from mixgrpo_amd.ops import Rows\n"
        "g = torch.Generator().manual_seed(3)\n"
        "M, N, K = 256 * 40 + 19, 2304, 448\n"
        "A = (torch.randn(M, K, generator=g) * 0.5).bfloat16().cuda(); W = (torch.randn(N, K, generator=g) * 0.1).bfloat16().cuda()\n"
        "b = (torch.randn(N, generator=g) * 0.2).bfloat16().cuda(); C = torch.empty(M, N, dtype=torch.bfloat16, device='cuda')\n"
        "ops.gemm(Rows.of(A), W, b, Rows.of(C), N, K, 0); torch.cuda.synchronize()\n"
        "ref = (A.float() @ W.float().t() + b.float())\n"
        "d = (C.float() - ref).abs(); tol = ref.abs() * 2.0 ** -7 + 2e-3\n"
        "assert (d <= tol).all(), (d - tol).max().item()\n"
        "print('OK')\n")
    for mode in ("0", None):
        env = dict(os.environ)
        env.pop("MGX_GEMM_MODE", None)
        if mode:
            env["MGX_GEMM_MODE"] = mode
        r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=300,
                           cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
        assert r.returncode == 0 and "OK" in r.stdout, (mode, r.stdout[-500:], r.stderr[-2000:])


@pytest.mark.parametrize("M,N,K", [(36864, 3072, 3072), (8192 + 256 * 3, 3072, 15360), (32768, 9216, 3072)])
def test_gemm_every_element_exact_on_integer_operands(M, N, K):
    """The ping-pong K-loop's LDS hazards (DMA stage reuse, the half-a-barrier lag of waves 4-7, the pipeline running on
    across output tiles) are a matter of construction; this is the screen for it: operands whose products and partial sums
    are exactly representable (multiples of 1/32 below 2^24 / 32), so EVERY output element of a FLUX-size GEMM has to equal
    the fp32 reference bit for bit whatever the summation order -- several tiles per workgroup, 48 and 240 K-tiles, repeated
    launches (a stage overwritten early or read late would show as a wrong element in some launch)."""
    from mixgrpo_amd import ops
    from mixgrpo_amd.ops import Rows
    g = torch.Generator(device="cuda").manual_seed(M + K)
    A = (torch.randint(-4, 5, (M, K), generator=g, device="cuda").float() / 8).bfloat16()
    W = (torch.randint(-2, 3, (N, K), generator=g, device="cuda").float() / 4).bfloat16()
    ref = torch.empty(M, N, dtype=torch.float32, device="cuda")
    for r0 in range(0, M, 8192):                                   # reference in row blocks (fp32 copies of A are large)
        ref[r0:r0 + 8192] = A[r0:r0 + 8192].float() @ W.float().t()
    C = torch.empty(M, N, dtype=torch.float32, device="cuda")
    for _ in range(4):
        C.fill_(float("nan"))
        ops.gemm(Rows.of(A), W, None, Rows.of(C), N, K, EPI_F32_ACC, beta=0.0)
        torch.cuda.synchronize()
        assert torch.equal(C, ref)


@pytest.mark.parametrize("N", [256, 64])          # 256: the persistent 256x256 kernel's register epilogue; 64: the 128x128 kernel's
def test_gelu_rows_is_the_gemm_epilogue_on_every_bf16_value(N):
    """`mgx_gelu_bf16` re-creates the bias+GELU epilogue's activation from a kept pre-activation (the recompute pass then
    skips the GEMM): the two have to agree BIT FOR BIT.  Exhaustive: a GEMM whose Linear output is exactly the value v
    (A[m][0] = v, W[n][0] = 1, everything else 0) for all 65536 bf16 patterns."""
    from mixgrpo_amd import ops
    from mixgrpo_amd.ops import Rows
    M, K = 65536, 128
    vals = torch.arange(65536, dtype=torch.int32).to(torch.int16).view(torch.bfloat16).cuda()
    finite = torch.isfinite(vals.float())
    A = torch.zeros(M, K, dtype=torch.bfloat16, device="cuda")
    A[:, 0] = torch.where(finite, vals, torch.zeros_like(vals))               # (inf * 0 would poison the row)
    W = torch.zeros(N, K, dtype=torch.bfloat16, device="cuda")
    W[:, 0] = 1.0
    C = torch.empty(M, N, dtype=torch.bfloat16, device="cuda")
    pre = torch.empty(M, N, dtype=torch.bfloat16, device="cuda")
    ops.gemm(Rows.of(A), W, None, Rows.of(C), N, K, EPI_GELU, aux=pre)
    assert torch.equal(pre[:, 0].view(torch.int16), A[:, 0].view(torch.int16)) or torch.equal(pre[:, 0].float(), A[:, 0].float())
    out = torch.full((M, N + 8), 3.0, dtype=torch.bfloat16, device="cuda")    # strided destination (the single block's cat)
    ops.gelu_rows(pre, N, out, N + 8, M, N)
    torch.cuda.synchronize()
    assert torch.equal(out[:, :N].view(torch.int16), C.view(torch.int16))     # bit for bit, every value, every column
    assert (out[:, N:] == 3.0).all()
    x = A[:, 0].float()
    big = x.abs() > 1e-30                                                     # (bf16 subnormals halve with a visible rounding)
    # (sanity against torch: its tanh saturates to exactly -1 below x ~ -5.5 where the true value is still ~ -1e-8)
    assert torch.allclose(C[:, 0].float()[big], torch.nn.functional.gelu(x[big], approximate="tanh"), rtol=2.0 ** -7, atol=1e-6)


# ---- stream-K tail (csrc/gemm.hip, sk_tail / gemm_sk_fixup_kernel): shapes whose last round is split along K under the
# default cost rule.  (tiles, per-XCD tail R, parts P = min(8, 32 // R)): 128 tiles = no whole round, R = 16, every tile in 2
# parts; 296 tiles = one whole round + R = 5, 6 parts (30 of an XCD's 32 workgroups busy); 356 tiles with a ragged M edge =
# one round + R = 13 on four XCDs and 12 on the others, 2 parts; 576 tiles = 2 rounds + R = 8, 4 parts.
SK_SHAPES = [(2048, 4096, 8192), (2048, 9472, 4096), (1024 - 17, 22784, 8192), (3072, 12288, 4096)]


def _sk_splits(M, N, K):
    """The cost rule of csrc/gemm.hip `launch()` restated, per XCD: T = 0.0247 K us; P = min(8, 32 // R);
    split <=> P >= max(2, floor(1.25 / (1 - 40 / T)) + 1) and K / 64 >= 4 P."""
    import math
    tiles = math.ceil(M / 256) * math.ceil(N / 256)
    room = 1 - 40.0 / (K * 0.0247)
    minparts = 9 if room <= 0 else max(2, math.floor(1.25 / room) + 1)
    q, rem = tiles // 8, tiles % 8
    out = []
    for cnt in [q + (x < rem) for x in range(8)]:
        R = cnt % 32
        P = min(8, 32 // R) if R else 0
        out.append(P >= 2 and P >= minparts and K // 64 >= 4 * P)
    return out


def test_stream_k_rule_splits_the_test_shapes():
    """The shapes of this file's stream-K tests ARE split -- otherwise they would silently test the unsplit kernel."""
    for shp in SK_SHAPES + [(2048, 4096, 8192 + 64), (3072, 12288, 4096 + 64), (6144, 3072, 4096)]:
        assert all(_sk_splits(*shp)), shp
    assert not any(_sk_splits(4608, 3072, 15360))       # R = 27: cannot be cut in two, left whole
    assert not any(_sk_splits(3072, 3072, 28672))       # R = 18 likewise


@pytest.mark.parametrize("M,N,K", SK_SHAPES)
@pytest.mark.parametrize("epi", [EPI_BIAS, EPI_GELU, EPI_GATE_RES, EPI_DGELU])
def test_gemm_stream_k_tail(M, N, K, epi):
    """The split launch agrees with the fp32 torch reference (same bar as the unsplit kernel), is deterministic, and agrees
    with the unsplit launch of the same problem to fp32-summation-order accuracy."""
    from mixgrpo_amd import ops
    from mixgrpo_amd.ops import Rows
    g = torch.Generator().manual_seed(M + N + K + epi)
    rpb = 1024 if epi == EPI_GATE_RES else 1 << 40
    if epi == EPI_GATE_RES:
        _, A_rows, A = _batched(M, K, rpb, 3, torch.bfloat16, g, 0.5)
        C_store, C_rows, C0 = _batched(M, N, rpb, 5, torch.bfloat16, g, 1.0)
    else:
        A = (torch.randn(M, K, generator=g) * 0.5).bfloat16().cuda()
        A_rows = Rows.of(A)
        C_store = torch.randn(M, N, generator=g).bfloat16().cuda()
        C_rows, C0 = Rows.of(C_store), C_store
    init = C_store.clone()
    W = (torch.randn(N, K, generator=g) * 0.03).bfloat16().cuda()
    bias = (torch.randn(N, generator=g) * 0.2).bfloat16().cuda()
    nb = (M + rpb - 1) // rpb if epi == EPI_GATE_RES else 1
    gate = torch.randn(nb, N, generator=g).bfloat16().cuda() if epi == EPI_GATE_RES else None
    aux = None
    if epi in (EPI_GELU, EPI_GATE_RES):
        aux = torch.full((M, N), 7.0, dtype=torch.bfloat16, device="cuda")
    elif epi == EPI_DGELU:
        aux = torch.randn(M, N, generator=g).bfloat16().cuda()
    outs = []
    assert ops.GEMM_STREAM_K
    for sk in (True, True, False):
        C_store.copy_(init)
        ops.GEMM_STREAM_K = sk
        try:
            ops.gemm(A_rows, W, bias, C_rows, N, K, epi, gate=gate, gate_ld=N, aux=aux)
        finally:
            ops.GEMM_STREAM_K = True
        torch.cuda.synchronize()
        outs.append((C_store.clone(), None if aux is None or epi == EPI_DGELU else aux.clone()))
    assert torch.equal(outs[0][0], outs[1][0])                                   # deterministic
    y = (A.float() @ W.float().t() + bias.float()).bfloat16().float()
    amp = 0.0
    if epi == EPI_BIAS:
        ref = y
    elif epi == EPI_GELU:
        ref, amp = _gelu(y), 1.2
    elif epi == EPI_GATE_RES:
        gsel = gate.float()[torch.arange(M, device="cuda") // rpb]
        ref, amp = C0.float() * 0 + init[:, :rpb].reshape(-1, N)[:M].float() + (gsel * y).bfloat16().float(), gsel.abs() + 0.01
    else:
        ref, amp = y * _dgelu(aux.float().cpu()).cuda(), 1.2
    pick = (lambda t: t[:, :rpb].reshape(-1, N)[:M]) if epi == EPI_GATE_RES else (lambda t: t)
    for C_out, aux_out in (outs[0], outs[2]):
        _close_bf16(pick(C_out), ref.bfloat16(), y, amp)
        if aux_out is not None:
            _close_bf16(aux_out, y.bfloat16())
    if epi == EPI_GATE_RES:
        assert torch.equal(outs[0][0][:, rpb:], init[:, rpb:])                   # padding rows between batches untouched
    # split vs unsplit: the same products, summed in a different order -- a handful of bf16 roundings may flip
    assert (pick(outs[0][0]) != pick(outs[2][0])).float().mean().item() < 2e-3


@pytest.mark.parametrize("M,N,K,beta", [(2048, 4096, 8192 + 64, 1.0), (3072, 12288, 4096 + 64, 1.0), (6144, 3072, 4096, 0.0)])
def test_gemm_stream_k_f32_accumulate(M, N, K, beta):
    """wgrad form with a split tail: 128 tiles in 2 parts, 576 (the 3072 x 12288 weight gradients) in 4 parts of 16.25 K-tiles,
    288 in 8 parts."""
    from mixgrpo_amd import ops
    from mixgrpo_amd.ops import Rows
    g = torch.Generator().manual_seed(13)
    A = (torch.randn(M, K, generator=g) * 0.5).bfloat16().cuda()
    W = (torch.randn(N, K, generator=g) * 0.5).bfloat16().cuda()
    C0 = torch.randn(M, N, generator=g).cuda()
    ref = beta * C0.double() + A.float().double() @ W.float().double().t()
    outs = []
    for _ in range(2):
        C = C0.clone()
        ops.gemm(Rows.of(A), W, None, Rows.of(C), N, K, EPI_F32_ACC, beta=beta)
        torch.cuda.synchronize()
        outs.append(C)
    assert torch.equal(outs[0], outs[1])
    err = ((outs[0].double() - ref).norm() / ref.norm()).item()
    assert err < 1e-5, err


# ---- pair launches (mgx_gemm_bf16_pair): the text- and image-stream Linear of a double block as one walk of the persistent kernel
@pytest.mark.parametrize("epi", [EPI_BIAS, EPI_GELU, EPI_GATE_RES, EPI_DGELU])
@pytest.mark.parametrize("B,L,Nimg,N,K,swap_w", [(4, 512, 1024, 3072, 1024, False), (2, 512, 4096, 1536, 512, True), (3, 256, 2048 + 64, 2048, 256, False),
                                                (4, 512, 1024, 3072, 4096, True)])    # the last: 288 tiles, K = 4096 -> pair AND stream-K tail
def test_gemm_pair_equals_two_launches(epi, B, L, Nimg, N, K, swap_w):
    """Joint [B, S, .] buffers with the text rows first (row-batched operands, as flux.py hands them over) or stacked plain
    matrices; weights in either address order.  With the stream-K tail off every tile is computed whole by the same K-loop in
    both forms: the pair launch must equal the two single launches BIT FOR BIT; with it on, to summation-order accuracy."""
    from mixgrpo_amd import ops
    from mixgrpo_amd.ops import Rows
    g = torch.Generator().manual_seed(B * 1000 + N + K + epi)
    S = L + Nimg
    joint_a = epi in (EPI_GATE_RES,)                         # A from a joint [B, S, K] buffer (to_out: the attention output)
    Wboth = (torch.randn(2, N, K, generator=g) * 0.05).bfloat16().cuda()
    Wt, Wi = (Wboth[1], Wboth[0]) if swap_w else (Wboth[0], Wboth[1])
    bias = (torch.randn(2, N, generator=g) * 0.2).bfloat16().cuda()
    if joint_a:
        Abuf = (torch.randn(B, S, K, generator=g) * 0.5).bfloat16().cuda()
        A1, A2 = Rows(Abuf, B * L, K, L, S * K), Rows(Abuf[0, L:], B * Nimg, K, Nimg, S * K)
    else:
        Abuf = (torch.randn(B * S, K, generator=g) * 0.5).bfloat16().cuda()
        A1, A2 = Rows.of(Abuf[:B * L]), Rows.of(Abuf[B * L:])
    C0 = torch.randn(B, S, N, generator=g).bfloat16().cuda()
    gate = torch.randn(2, B, N, generator=g).bfloat16().cuda() if epi == EPI_GATE_RES else None
    aux0 = torch.randn(B * S, N, generator=g).bfloat16().cuda() if epi == EPI_DGELU else \
        (torch.full((B * S, N), 7.0, dtype=torch.bfloat16, device="cuda") if epi in (EPI_GELU, EPI_GATE_RES) else None)

    pair_default = ops.GEMM_PAIR

    def run(pair, sk):
        C = C0.clone()
        aux = None if aux0 is None else aux0.clone()
        if epi == EPI_GATE_RES:                              # C joint [B, S, N], text rows first
            C1, C2 = Rows(C, B * L, N, L, S * N), Rows(C[0, L:], B * Nimg, N, Nimg, S * N)
        else:                                                # C stacked
            Cs = C.view(B * S, N)
            C1, C2 = Rows.of(Cs[:B * L]), Rows.of(Cs[B * L:])
        a1 = None if aux is None else aux[:B * L]
        a2 = None if aux is None else aux[B * L:]
        g1 = None if gate is None else gate[0]
        g2 = None if gate is None else gate[1]
        ops.GEMM_STREAM_K, ops.GEMM_PAIR = sk, pair
        try:
            ops.gemm_pair(A1, Wt, bias[0], C1, A2, Wi, bias[1], C2, N, K, epi, gate1=g1, gate2=g2, gate_ld=N, aux1=a1, aux2=a2)
        finally:
            ops.GEMM_STREAM_K, ops.GEMM_PAIR = True, pair_default
        torch.cuda.synchronize()
        return C, aux

    Cp, ap = run(True, False)
    Cs_, as_ = run(False, False)
    assert torch.equal(Cp, Cs_)
    if aux0 is not None and epi != EPI_DGELU:
        assert torch.equal(ap, as_)
    Ck, _ = run(True, True)
    assert (Ck != Cs_).float().mean().item() < 2e-3
    # and against the fp32 reference, per stream
    Ad = Abuf.view(B, S, K) if joint_a else None
    for which, (W_, sl_) in enumerate(((Wt, slice(0, L)), (Wi, slice(L, S)))):
        if joint_a:
            a = Ad[:, sl_].reshape(-1, K)
        else:
            a = (Abuf[:B * L] if which == 0 else Abuf[B * L:])
        y = (a.float() @ W_.float().t() + bias[which].float()).bfloat16().float()
        if epi == EPI_GATE_RES:
            rows = sl_.stop - sl_.start
            gsel = gate[which].float().repeat_interleave(rows, dim=0)
            ref = C0[:, sl_].reshape(-1, N).float() + (gsel * y).bfloat16().float()
            out, amp = Cp[:, sl_].reshape(-1, N), gsel.abs() + 0.01
        else:
            out = Cp.view(B * S, N)[:B * L] if which == 0 else Cp.view(B * S, N)[B * L:]
            if epi == EPI_BIAS:
                ref, amp = y, 0.0
            elif epi == EPI_GELU:
                ref, amp = _gelu(y), 1.2
            else:
                au = aux0[:B * L] if which == 0 else aux0[B * L:]
                ref, amp = y * _dgelu(au.float().cpu()).cuda(), 1.2
        _close_bf16(out, ref.bfloat16(), y, amp)


@pytest.mark.parametrize("M,N", [(4608, 12288), (1000, 256), (64 * 3 + 8, 136)])
def test_transpose_with_gelu_equals_gelu_then_transpose(M, N):
    """`mgx_transpose_gelu_bf16` (the weight-gradient operand of ff.net.2 / proj_out formed from a KEPT pre-activation) must be
    bit for bit the transpose of what `mgx_gelu_bf16` writes -- which tests above hold to the GEMM epilogue's activation."""
    from mixgrpo_amd import ops
    from mixgrpo_amd.ops import Rows
    g = torch.Generator().manual_seed(M + N)
    x = (3 * torch.randn(M, N, generator=g)).bfloat16().cuda()
    Mp = (M + 63) // 64 * 64
    y = torch.empty_like(x)
    ops.gelu_rows(x, N, y, N, M, N)
    want = torch.zeros(N, Mp, dtype=torch.bfloat16, device="cuda")
    ops.transpose(Rows.of(y), N, want, Mp)
    got = torch.full((N, Mp), 7.0, dtype=torch.bfloat16, device="cuda")
    ops.transpose(Rows.of(x), N, got, Mp, gelu=True)
    torch.cuda.synchronize()
    assert torch.equal(got, want)
    ref = torch.nn.functional.gelu(x.float(), approximate="tanh")
    assert ((got[:, :M].t().float() - ref).abs() <= ref.abs() * 2.0 ** -7 + 1e-3).all()
    assert (got[:, M:] == 0).all()


@pytest.mark.parametrize("B,rows,F,K,s0,Sp", [(8, 512, 3072, 1024, 0, 4608), (4, 2048, 1536, 512, 512, 2560), (1, 36864 // 8, 3072, 256, 0, 4608),
                                              (3, 1024 + 64, 3072, 4096, 64, 1216)])
@pytest.mark.parametrize("sk", [False, True])
def test_linear_t_writes_the_transposed_projection(B, rows, F, K, s0, Sp, sk):
    """mgx_linear_bf16_t (the value projection landing as V^T [B, F, Sp] at sequence offset s0: operand roles swapped, row-wise
    bias, column batches) against the plain Linear of the same operands, transposed on the host.  Without the stream-K tail
    both forms run every output element's K-loop whole and in the same order: equal BIT FOR BIT; with it, to summation-order
    accuracy (one bf16 ulp on isolated elements).  Everything outside columns s0 .. s0 + rows stays untouched."""
    from mixgrpo_amd import ops
    from mixgrpo_amd.ops import Rows
    g = torch.Generator().manual_seed(B + rows + F + K)
    tokens = B * rows
    X = (torch.randn(tokens, K, generator=g) * 0.5).bfloat16().cuda()
    W = (torch.randn(F, K, generator=g) * 0.05).bfloat16().cuda()
    bias = (torch.randn(F, generator=g) * 0.3).bfloat16().cuda()
    sk_default = ops.GEMM_STREAM_K
    ops.GEMM_STREAM_K = sk
    try:
        Ct = torch.full((B, F, Sp), 7.0, dtype=torch.bfloat16, device="cuda")
        assert ops.linear_t(X, W, bias, Ct.view(-1)[s0:], tokens, F, K, Sp, rows, F * Sp)
        C = torch.empty(tokens, F, dtype=torch.bfloat16, device="cuda")
        ops.gemm(Rows.of(X), W, bias, Rows.of(C), F, K)
    finally:
        ops.GEMM_STREAM_K = sk_default
    want = C.view(B, rows, F).transpose(1, 2)
    got = Ct[:, :, s0:s0 + rows]
    assert (Ct[:, :, :s0] == 7.0).all() and (Ct[:, :, s0 + rows:] == 7.0).all()
    ref = (X.float() @ W.float().t() + bias.float()).view(B, rows, F).transpose(1, 2)
    assert ((got.float() - ref).norm() / ref.norm()).item() < 3e-3
    if not sk:
        assert torch.equal(got, want)
    else:
        assert ((got.float() - want.float()).abs() <= 0.0079 * want.float().abs() + 1e-6).all()     # <= 1 bf16 ulp
        assert (got != want).float().mean().item() < 1e-2


def test_linear_t_declines_what_the_persistent_kernel_cannot_take():
    """Fewer than 128 output tiles / token batches that are no multiple of 64: returns False and writes nothing (the caller keeps
    the plain projection and the transposing pass)."""
    from mixgrpo_amd import ops
    X = torch.zeros(512, 256, dtype=torch.bfloat16, device="cuda")
    W = torch.zeros(3072, 256, dtype=torch.bfloat16, device="cuda")
    Ct = torch.full((1, 3072, 512), 7.0, dtype=torch.bfloat16, device="cuda")
    assert not ops.linear_t(X, W, None, Ct, 512, 3072, 256, 512, 512, 3072 * 512)            # 24 tiles
    X = torch.zeros(8 * 1000, 256, dtype=torch.bfloat16, device="cuda")
    Ct = torch.full((8, 3072, 1024), 7.0, dtype=torch.bfloat16, device="cuda")
    assert not ops.linear_t(X, W, None, Ct, 8000, 3072, 256, 1024, 1000, 3072 * 1024)        # 1000 % 64 != 0
    assert (Ct == 7.0).all()


@pytest.mark.parametrize("B,rows,H,K,s0,S", [(8, 512, 24, 1024, 0, 4608), (4, 2176, 4, 512, 128, 2432), (6, 1280, 6, 256, 0, 1280),
                                             (1, 4736, 8, 320, 64, 4800)])
@pytest.mark.parametrize("q_scale,paired", [(1.0, False), (1.4426950408889634 / 128 ** 0.5, False), (1.4426950408889634 / 128 ** 0.5, True)])
def test_linear_qk_norm_rope_equals_projection_then_norm_pass(B, rows, H, K, s0, S, q_scale, paired):
    """mgx_linear_qk_norm_rope (RMSNorm + RoPE + head split in the q | k projection's epilogue: sums of squares crossing two
    waves through LDS, the K-loop's early / late barrier protocol kept) against mgx_gemm_bf16 + mgx_qk_norm_rope_fwd_qs on the
    same operands: the epilogue restates that kernel's arithmetic in its summation order, so Q and K are equal BIT FOR BIT;
    positions outside s0 .. s0 + rows stay untouched.  (4736 rows: the last tile row is half outside the matrix.)"""
    from mixgrpo_amd import ops
    from mixgrpo_amd.ops import Rows
    g = torch.Generator().manual_seed(B * 7 + rows + H + K)
    d, tokens = H * 128, B * rows
    X = (torch.randn(tokens, K, generator=g) * 0.7).bfloat16().cuda()
    W = (torch.randn(3 * d, K, generator=g) * 0.06).bfloat16().cuda()
    bias = (torch.randn(3 * d, generator=g) * 0.3).bfloat16().cuda()
    wq = (1 + 0.2 * torch.randn(128, generator=g)).cuda()
    wk = (1 + 0.2 * torch.randn(128, generator=g)).cuda()
    ang = torch.rand(S, 64, generator=g) * 6.28
    cos = torch.cos(ang).repeat_interleave(2, dim=1).contiguous().cuda()
    sin = torch.sin(ang).repeat_interleave(2, dim=1)
    if not paired:                                   # general tables: the two entries of a pair differ
        sin = sin * (1 + 0.01 * torch.randn(S, 128, generator=g))
    sin = sin.contiguous().cuda()
    pairs = ops.rope_pair_table(cos, sin)
    assert (pairs is not None) == paired
    Sp = (S + 63) // 64 * 64
    Q1 = torch.full((B, H, S, 128), 7.0, dtype=torch.bfloat16, device="cuda")
    K1 = torch.full_like(Q1, 7.0)
    assert ops.linear_qk_norm_rope(X, W[:2 * d], bias[:2 * d], wq, wk, cos, sin, Q1, K1, B, H, S, rows, s0, K, q_scale=q_scale,
                                   pairs=pairs)
    qkv = torch.zeros(tokens, 3 * d, dtype=torch.bfloat16, device="cuda")
    sk_default = ops.GEMM_STREAM_K
    ops.GEMM_STREAM_K = False
    try:
        ops.gemm(Rows.of(X), W[:2 * d], bias[:2 * d], Rows(qkv, tokens, 3 * d), 2 * d, K)
    finally:
        ops.GEMM_STREAM_K = sk_default
    Q0 = torch.full_like(Q1, 7.0)
    K0 = torch.full_like(Q1, 7.0)
    ops.qk_norm_rope(qkv, wq, wk, cos, sin, Q0, K0, None, B, H, S, Sp, rows, s0, q_scale=q_scale)
    assert (Q0[:, :, s0:s0 + rows] != 7.0).any()
    assert torch.equal(Q1, Q0) and torch.equal(K1, K0)


def test_linear_qk_norm_rope_declines():
    from mixgrpo_amd import ops
    z = lambda *s: torch.zeros(*s, dtype=torch.bfloat16, device="cuda")
    f = lambda *s: torch.zeros(*s, device="cuda")
    # 3 heads (d = 384, not a multiple of 256) / rows % 128 != 0 / too few tiles
    assert not ops.linear_qk_norm_rope(z(8192, 256), z(768, 256), z(768), f(128), f(128), f(1024, 128), f(1024, 128), z(8, 3, 1024, 128),
                                       z(8, 3, 1024, 128), 8, 3, 1024, 1024, 0, 256)
    assert not ops.linear_qk_norm_rope(z(8000, 256), z(1024, 256), z(1024), f(128), f(128), f(1000, 128), f(1000, 128), z(8, 4, 1000, 128),
                                       z(8, 4, 1000, 128), 8, 4, 1000, 1000, 0, 256)
    assert not ops.linear_qk_norm_rope(z(1024, 256), z(1024, 256), z(1024), f(128), f(128), f(1024, 128), f(1024, 128), z(1, 4, 1024, 128),
                                       z(1, 4, 1024, 128), 1, 4, 1024, 1024, 0, 256)
