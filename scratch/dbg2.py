import torch, sys
sys.path.insert(0, "tests")
from helpers import load_golden
T_, _ = load_golden("solver_steps")
s = (3.0*torch.linspace(1,0,26))/(1+2.0*torch.linspace(1,0,26))
print("sigma diff idx:", (s != T_["sigma/shift3.0_T25"]).nonzero().flatten().tolist())
print(torch.__config__.show().split("CPU capability")[1][:60] if "CPU capability" in torch.__config__.show() else "")
print(torch.backends.cpu.get_cpu_capability())
