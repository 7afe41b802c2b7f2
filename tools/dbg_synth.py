import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dl_attack_on_imagenet_amd import ops
dev = torch.device("cuda")
for (b, k, dt, P) in ((96, 128, torch.float32, 150528), (96, 128, torch.float32, 150528), (96, 112, torch.float32, 150528), (96, 128, torch.float32, 32768), (32, 128, torch.float32, 150528), (96, 128, torch.bfloat16, 150528)):
    gen = torch.Generator().manual_seed(b + k)
    d = (-1 + 2 * torch.rand(1, 1, P, k, generator=gen)).to(dev)
    v = (torch.randn(b, k, generator=gen) * 0.02).to(dev)
    x = torch.rand(b, 1, 1, P, generator=gen).to(dev).to(dt)
    vp = ops.pack_codes(v, None, b)
    dq, vq = (d.bfloat16().double(), v.bfloat16().double()) if dt == torch.bfloat16 else (d.double(), v.double())
    ref = (x[:b].double().reshape(b, -1) + vq @ dq.reshape(-1, k).t())
    for rep in range(2):
        out = ops.synth(x[:b], d, vp, b)
        err = (out.double().reshape(b, -1) - ref).abs()
        bad = (err > 1e-2).nonzero()
        print(b, k, dt, P, "max err", float(err.max()), "nbad", bad.shape[0])
        if bad.shape[0]:
            rows = bad[:, 0].unique(); cols = bad[:, 1].unique()
            print("  bad rows", rows.tolist(), "\n  bad tiles", (cols // 128).unique().tolist(), "\n  cols%128 uniq", (cols % 128).unique().tolist())
            r0, c0 = bad[0].tolist()
            print("  sample", r0, c0, float(out.reshape(b, -1)[r0, c0]), float(ref[r0, c0]), float(x.reshape(b, -1)[r0, c0]))
print("---- which row's value landed in the bad slots?")
b, k, dt, P = 96, 128, torch.float32, 150528
gen = torch.Generator().manual_seed(b + k)
d = (-1 + 2 * torch.rand(1, 1, P, k, generator=gen)).to(dev)
v = (torch.randn(b, k, generator=gen) * 0.02).to(dev)
x = torch.rand(b, 1, 1, P, generator=gen).to(dev).to(dt)
vp = ops.pack_codes(v, None, b)
ref = (x.double().reshape(b, -1) + v.double() @ d.double().reshape(-1, k).t())
for rep in range(3):
    out = ops.synth(x, d, vp, b).reshape(b, -1).double()
    bad = ((out - ref).abs() > 1e-2).nonzero()
    from collections import Counter
    cnt = Counter()
    for r0, c0 in bad[:400].tolist():
        col = ref[:, c0 - 3:c0 + 4]
        m = ((col - out[r0, c0]).abs() < 1e-5).nonzero()
        cnt[tuple((int(a) - r0, int(bb) - 3) for a, bb in m.tolist())] += 1
    print(bad.shape[0], cnt.most_common(8))
