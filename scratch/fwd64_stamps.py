"""In-kernel shader-clock stamps of attn_fwd64 (diagnostic harness scratch/fwd64_diag.hip): where a workgroup's cycles go
(prologue / first iteration / steady-state loop / last iteration / epilogue) and the clock the chip holds."""
import ctypes as C, math, os, subprocess, sys, torch
sys.path.insert(0, ".")
variant = sys.argv[1] if len(sys.argv) > 1 and sys.argv[1] != "-" else ""
acc = len(sys.argv) > 2 and sys.argv[2] == "acc"          # the accumulator-initialised stream (attn_fwd64q) on a prescaled Q
subprocess.check_call([sys.executable, "mixgrpo_amd/csrc/gen/attn_fwd64.py", "--diag"] + (["--acc"] if acc else []) +
                      ([f"--timing-only={variant}"] if variant else []))
print(f"==== variant: {variant or 'real kernel'}{' (attn_fwd64q)' if acc else ''}")
so = f"scratch/libfwd64_diag_{variant.replace(',', '_') or 'real'}{'_acc' if acc else ''}.so"
subprocess.check_call(["hipcc", "-O3", "-std=c++17", "-fPIC", "-shared", "--offload-arch=gfx950", "-Iscratch"] + (["-DDIAG_ACC"] if acc else []) +
                      ["scratch/fwd64_diag.hip", "-o", so])
lib = C.CDLL(so)
lib.fwd64_diag.argtypes = [C.c_void_p] * 5 + [C.c_int] * 3 + [C.c_long, C.c_long, C.c_float, C.c_void_p]
B, H, S = 8, 24, 4608
torch.manual_seed(0)
q, k, v = (torch.randn(B, H, S, 128, device="cuda").bfloat16() for _ in range(3))
if acc:
    q = (q.float() * (1.4426950408889634 / math.sqrt(128))).bfloat16()
vt = v.transpose(-1, -2).contiguous()
O = torch.empty(B, S, H * 128, device="cuda", dtype=torch.bfloat16)
nwg = (S // 256) * H * B
dbg = torch.zeros(nwg * 32, dtype=torch.int64, device="cuda")
st = torch.cuda.current_stream().cuda_stream
def run():
    assert lib.fwd64_diag(q.data_ptr(), k.data_ptr(), vt.data_ptr(), O.data_ptr(), dbg.data_ptr(), B, H, S, H * 128, S * H * 128,
                          1 / math.sqrt(128), st) == 0
for _ in range(200): run()          # ~0.4 s of back-to-back launches: the clock has settled
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(); run(); e1.record(); torch.cuda.synchronize()
print(f"kernel {e0.elapsed_time(e1):.3f} ms")
d = dbg.view(nwg, 4, 8).cpu().double()
t = d[:, :, :6]
seg = (t[:, :, 1:] - t[:, :, :-1])
names = ["prologue", "iteration 0", f"loop ({S // 64 - 2} iterations)", "last iteration + tail", "epilogue"]
med = seg.median(dim=0).values.median(dim=0).values
tot = (t[:, :, 5] - t[:, :, 0]).median().item()
for n, c in zip(names, med.tolist()):
    print(f"{n:32s} {c:10.0f} cycles  {100 * c / tot:5.1f} %")
print(f"total per workgroup {tot:.0f} cycles; per steady iteration {med[2].item() / (S // 64 - 2):.1f} cycles (MFMA floor 2048)")
# clock: first and last workgroup records of one wave on the same CU would be ideal; use total span per WG vs realtime is not
# stamped at entry, so estimate from kernel time: cycles per WG * rounds / kernel time
rounds = math.ceil(nwg / 256)
print(f"implied clock if the 256 CUs ran {rounds} rounds back to back: {tot * rounds / (e0.elapsed_time(e1) * 1e-3) / 1e9:.2f} GHz")
