"""GPU experiment (not product): what bounds the frozen classifier's fwd + input-gradient time."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.nn as nn, torch.nn.functional as F
from dl_attack_on_imagenet_amd import zoo

dev = torch.device("cuda")
B = int(os.environ.get("B", 512))

def timeit(fn, n=5, w=2):
    for _ in range(w): fn()
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t) / n * 1e3

def fold_bn(module):
    """Fold eval-mode BatchNorm2d into the preceding Conv2d, in place, for Sequential-like parents."""
    for name, child in list(module.named_children()):
        fold_bn(child)
    names = list(module._modules.keys())
    # pattern: attributes convN / bnN in resnet blocks, or Sequential [conv, bn]
    pairs = []
    if isinstance(module, nn.Sequential):
        for a, b in zip(names, names[1:]):
            if isinstance(module._modules[a], nn.Conv2d) and isinstance(module._modules[b], nn.BatchNorm2d):
                pairs.append((a, b))
    else:
        for a in names:
            if a.startswith("conv") and isinstance(module._modules[a], nn.Conv2d):
                b = "bn" + a[4:]
                if b in module._modules and isinstance(module._modules[b], nn.BatchNorm2d):
                    pairs.append((a, b))
    for a, b in pairs:
        conv, bn = module._modules[a], module._modules[b]
        w = conv.weight.float(); scale = bn.weight.float() / torch.sqrt(bn.running_var.float() + bn.eps)
        bias = bn.bias.float() - bn.running_mean.float() * scale
        if conv.bias is not None: bias = bias + conv.bias.float() * scale
        new = nn.Conv2d(conv.in_channels, conv.out_channels, conv.kernel_size, conv.stride, conv.padding, conv.dilation, conv.groups, bias=True)
        new.weight.data = (w * scale.view(-1, 1, 1, 1)).to(conv.weight.dtype); new.bias.data = bias.to(conv.weight.dtype)
        new = new.to(conv.weight.device)
        if conv.weight.is_contiguous(memory_format=torch.channels_last): new = new.to(memory_format=torch.channels_last)
        for p in new.parameters(): p.requires_grad_(False)
        module._modules[a] = new; module._modules[b] = nn.Identity()
    return module

def run(model, x, labels):
    xt = x.detach().requires_grad_(True)
    out = model(xt)
    loss = F.cross_entropy(out.float(), labels, reduction="sum")
    (g,) = torch.autograd.grad(loss, xt)
    return g

x = torch.rand(B, 3, 224, 224, device=dev).bfloat16()
labels = torch.randint(0, 1000, (B,), device=dev)
res = {}
for cl in (1, 0):
    m = zoo.build_classifier("resnet50", device=dev, dtype=torch.bfloat16, channels_last=bool(cl))
    res[f"base cl={cl} fwd+bwd"] = timeit(lambda: run(m, x, labels))
    with torch.no_grad(): res[f"base cl={cl} fwd only"] = timeit(lambda: m(x))
    print(res, flush=True)
m = zoo.build_classifier("resnet50", device=dev, dtype=torch.bfloat16, channels_last=True)
g0 = run(m, x, labels)
torch.backends.cudnn.benchmark = True
res["cl=1 cudnn.benchmark fwd+bwd"] = timeit(lambda: run(m, x, labels)); print(res, flush=True)
torch.backends.cudnn.benchmark = False
fold_bn(m)
g1 = run(m, x, labels)
print("fold_bn grad rel diff", float((g1.float() - g0.float()).norm() / g0.float().norm()))
res["cl=1 foldbn fwd+bwd"] = timeit(lambda: run(m, x, labels))
with torch.no_grad(): res["cl=1 foldbn fwd only"] = timeit(lambda: m(x))
print(res, flush=True)
# conv1 backward-data alone: C_in = 3 vs padded
for cin in (3, 4, 8):
    conv = nn.Conv2d(cin, 64, 7, 2, 3, bias=False).to(dev).bfloat16().to(memory_format=torch.channels_last)
    for p in conv.parameters(): p.requires_grad_(False)
    xi = torch.rand(B, cin, 224, 224, device=dev).bfloat16().contiguous(memory_format=torch.channels_last).requires_grad_(True)
    y = conv(xi); gy = torch.rand_like(y)
    res[f"conv1 bwd-data cin={cin} NHWC"] = timeit(lambda: torch.autograd.grad(y, xi, gy, retain_graph=True))
    xi2 = torch.rand(B, cin, 224, 224, device=dev).bfloat16().requires_grad_(True)
    conv2 = nn.Conv2d(cin, 64, 7, 2, 3, bias=False).to(dev).bfloat16()
    for p in conv2.parameters(): p.requires_grad_(False)
    y2 = conv2(xi2); gy2 = torch.rand_like(y2)
    res[f"conv1 bwd-data cin={cin} NCHW"] = timeit(lambda: torch.autograd.grad(y2, xi2, gy2, retain_graph=True))
    res[f"conv1 fwd cin={cin} NHWC"] = timeit(lambda: conv(xi))
    print(res, flush=True)
# conv1 bwd-data in fp32 NCHW
conv3 = nn.Conv2d(3, 64, 7, 2, 3, bias=False).to(dev)
xi3 = torch.rand(B, 3, 224, 224, device=dev).requires_grad_(True); y3 = conv3(xi3); gy3 = torch.rand_like(y3)
res["conv1 bwd-data cin=3 fp32 NCHW"] = timeit(lambda: torch.autograd.grad(y3, xi3, gy3, retain_graph=True))
for k, v in res.items(): print(f"{k:45s} {v:9.2f} ms")
