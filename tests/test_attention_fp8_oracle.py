"""CPU checks of the fp8-attention oracle (oracle/attention_fp8.py): quantiser properties and the key permutation the
kernel's V^T operand uses.  (The reference has no fp8 attention: parity unpinned, see the oracle's header.)"""
import torch

from oracle import attention_fp8 as OA


def test_quantiser_properties():
    g = torch.Generator().manual_seed(1)
    x = (torch.randn(2, 3, 50, 128, generator=g) * 3).bfloat16()
    am = OA.amax_table(x, x, x)
    assert am.shape == (3, 6) and torch.equal(am[0], x.float().abs().reshape(6, -1).amax(1))
    q = OA.quantize(x, am[0])
    assert q.dtype == torch.float8_e4m3fn
    assert torch.equal(q.float().abs().reshape(6, -1).amax(1), torch.full((6,), 448.0))    # the amax element maps to 448
    deq = q.float() * (am[0] / 448.0).view(2, 3, 1, 1)
    big = x.float().abs() > am[0].view(2, 3, 1, 1) * 2.0 ** -6                              # well inside the normal range
    assert ((deq - x.float()).abs()[big] <= x.float().abs()[big] * 2.0 ** -4 * 1.01).all()  # 3 mantissa bits, RNE
    z = OA.quantize(torch.zeros(1, 1, 4, 128).bfloat16(), torch.zeros(1))
    assert torch.equal(z.view(torch.uint8), torch.zeros(1, 1, 4, 128, dtype=torch.uint8))


def test_key_order_is_the_accumulator_order():
    order = OA.key_order()
    assert sorted(order) == list(range(64))
    # lane half h holds, for S^T block kb, accumulator register i = key 32kb + 8(i>>2) + 4h + (i&3) (32x32 C/D layout)
    for h in range(2):
        for kb in range(2):
            for i in range(16):
                assert order[32 * h + 16 * kb + i] == 32 * kb + 8 * (i >> 2) + 4 * h + (i & 3)
    v8 = torch.arange(2 * 70 * 128, dtype=torch.int64).remainder(251).to(torch.uint8).view(1, 2, 70, 128).view(torch.float8_e4m3fn)
    vt = OA.v8t_layout(v8, 128)
    raw = v8.view(torch.uint8)
    assert vt.shape == (1, 2, 128, 128)
    for p in (0, 5, 17, 40, 63):
        key = order[p]
        assert torch.equal(vt[0, :, :, p], raw[0, :, key, :])
        if 64 + key < 70:
            assert torch.equal(vt[0, :, :, 64 + p], raw[0, :, 64 + key, :])
        else:
            assert (vt[0, :, :, 64 + p] == 0).all()


def test_attention_oracle_close_to_full_precision():
    g = torch.Generator().manual_seed(3)
    Q, K, V = (torch.randn(1, 2, 90, 128, generator=g).bfloat16() for _ in range(3))
    O, lse = OA.attention(Q, K, V)
    full = torch.softmax(torch.einsum("bhqd,bhkd->bhqk", Q.double(), K.double()) / 128 ** 0.5, -1) @ V.double()
    assert ((O - full).norm() / full.norm()).item() < 7e-2
    O2, _ = OA.attention(Q, K, V, quantize_p=True)
    assert 0 < ((O2 - O).norm() / O.norm()).item() < 4e-2
