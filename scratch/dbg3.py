import sys, torch
sys.path.insert(0, "tests"); sys.path.insert(0, "tests/golden"); sys.path.insert(0, ".")
from helpers import load_golden
from mixgrpo_amd import sampling_utils as SU
from oracle import solver as O
T_, M_ = load_golden("solver_steps")
x, v = T_["in/x"], T_["in/v"]
sig = T_["sigma/shift3.0_T8"]
k="dance/i0_sde0"
out = SU.dance_grpo_step(v.cuda(), x.cuda(), 0.3, sig, 0, None, True, False, noise=T_[k+"/noise"].cuda())
ora = O.dance_grpo_step(v, x, 0.3, sig, 0, None, True, False, noise=T_[k+"/noise"])
print("logp", out[2], ora[2], T_[k+"/logp"])
rp = SU.dance_grpo_step(v.cuda(), x.cuda(), 0.3, sig, 0, out[0].clone(), True, True)
print("rp", rp[2], T_[k+"/replay_logp_sde"])
kk = SU.dance_coeffs(sig, 0, 0.3); print([(n, getattr(kk,n)) for n,_ in kk._fields_])
