"""CPU check of the prototype stream (tests/asm_emu.py): one 256 x 256 tile, K = 256 / 384, against numpy, both visibility modes."""
import os, sys
import numpy as np
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", "..", "tests"))
sys.path.insert(0, HERE)
import asm_emu, gen

def bf16(x):
    u = np.ascontiguousarray(x, dtype=np.float32).view(np.uint32).astype(np.uint64)
    return ((u + 0x7FFF + ((u >> 16) & 1)) >> 16).astype(np.uint16)
def f32(h): return (h.astype(np.uint32) << 16).view(np.float32)

text = gen.generate()
haz = asm_emu.check_hazards(text)
print("hazards:", haz[:10], len(haz))
for K, mode, order in ((256, "late", [0, 1, 2, 3]), (512, "early", [3, 1, 0, 2]), (256, "early", [0, 1, 2, 3])):
    rng = np.random.default_rng(K)
    M = N = 256
    Mtot, Ntot = 512, 768                      # the tile sits inside larger matrices: m0 = 256, n0 = 512
    A = bf16(rng.standard_normal((Mtot, K)).astype(np.float32)); W = bf16(rng.standard_normal((Ntot, K)).astype(np.float32) * 0.1)
    C = np.zeros((Mtot, Ntot), np.uint16)
    m0, n0 = 256, 512
    inputs = dict(tid=np.arange(256).reshape(4, 64), a=("ptr", "A", m0 * K * 2), w=("ptr", "W", n0 * K * 2),
                  c=("ptr", "C", (m0 * Ntot + n0) * 2), lda2=K * 2, ldc2=Ntot * 2, nloop=K // 128, kmax=(K // 64 - 1) * 128)
    m = asm_emu.Machine(text, inputs, dict(A=A, W=W, C=C), lds_bytes=131072, mode=mode, order=order).run()
    ref = f32(A[m0:m0 + 256]).astype(np.float64) @ f32(W[n0:n0 + 256]).astype(np.float64).T
    got = f32(C[m0:m0 + 256, n0:n0 + 256]).astype(np.float64)
    rel = np.linalg.norm(got - ref) / np.linalg.norm(ref)
    outside = C.copy(); outside[m0:m0 + 256, n0:n0 + 256] = 0
    print(f"K {K} mode {mode}: rel {rel:.2e}  mfma {m.mfma_count}  stores outside tile: {int(outside.any())}")
    assert rel < 4e-3 and not outside.any() and m.mfma_count == 4 * 64 * (K // 64)
print("OK")
