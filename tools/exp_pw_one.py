"""One pointwise-forward shape in a loop (for rocprofv3 --pmc): python tools/exp_pw_one.py HW K N RES"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dl_attack_on_imagenet_amd import ops, _lib
dev = torch.device("cuda"); lib = _lib.load()
hw, k, n, has_res = (int(a) for a in sys.argv[1:5])
m = 512 * hw * hw
x = torch.randn(m, k, device=dev, dtype=torch.bfloat16); w = torch.randn(n, k, device=dev, dtype=torch.bfloat16) * 0.05
sc = torch.rand(n, device=dev) + 0.5; sh = torch.randn(n, device=dev)
r = torch.randn(m, n, device=dev, dtype=torch.bfloat16) if has_res else None
y = torch.empty(m, n, device=dev, dtype=torch.bfloat16)
for _ in range(10):
    lib.adil_pw_conv_fwd(ops._ptr(x), ops._ptr(w), ops._ptr(sc), ops._ptr(sh), ops._ptr(r), ops._ptr(y), m, k, n, 1, None, None, 0, 0, ops._stream())
torch.cuda.synchronize()
