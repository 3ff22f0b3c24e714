import sys, torch
sys.path.insert(0, "tests"); sys.path.insert(0, "tests/golden"); sys.path.insert(0, ".")
from helpers import load_golden
from mixgrpo_amd import sampling_utils as SU
T_, M_ = load_golden("solver_steps")
for case in [c for c in M_["cases"] if c["kind"] == "flow"]:
    k = case["key"]
    sig = SU.sd3_time_shift(3.0, torch.linspace(1, 0, case["T"] + 1))
    out = SU.flow_grpo_step(T_["in/v"].cuda(), T_["in/x"].cuda(), 0.7, sig, case["index"], None, determistic=case["det"], noise=T_[k+"/noise"].cuda())
    for nm, t in zip(("prev","x0","logp","mean"), out):
        d = (t.cpu() != T_[f"{k}/{nm}"])
        if nm != "logp" and d.any():
            idx = d.nonzero()[0]
            print(k, nm, int(d.sum()), "of", d.numel(), t.cpu()[tuple(idx)].item(), T_[f"{k}/{nm}"][tuple(idx)].item(),
                  "x", T_["in/x"][tuple(idx)].item(), "v", T_["in/v"][tuple(idx)].item())
    kk = SU.flow_coeffs(sig, case["index"], 0.7)
    if case["index"] == 2 and case["T"] == 25: print([ (n, getattr(kk, n)) for n,_ in kk._fields_])
