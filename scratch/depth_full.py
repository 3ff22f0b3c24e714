"""One-off evidence run: tests/test_hip_mmdit.py::test_full_width_error_vs_depth at FLUX.1-dev's full depth (19 + 38 blocks,
11.9 B parameters, d = 3072; 64 image + 32 text tokens) -- forward rel-L2 and parameter-gradient cosine of the HIP MMDiT
against the CPU oracle.  Too slow / large for the test suite (fp32 oracle weights + autograd on the host: ~150 GB of host
memory); writes its row into gpurun_out/r03_depth_parity.json like the suite's 1+1 / 2+2 / 4+8 cases."""
import json, os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
torch.set_num_threads(16)
import test_hip_mmdit as T
t0 = time.time()
try:
    T.test_full_width_error_vs_depth(19, 38)
    print("within the suite's bounds", flush=True)
except AssertionError as e:
    print("outside the suite's bounds:", e, flush=True)
print(json.load(open("gpurun_out/r03_depth_parity.json")).get("fwd_bwd_19+38"), f"{time.time() - t0:.0f} s", flush=True)
