"""fp32 plain ResNet-50, 512 images: forward + input-gradient backward, NCHW vs channels_last (the long leg of the ASR test)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from dl_attack_on_imagenet_amd import engine, zoo
dev = "cuda"
x = torch.rand(512, 3, 224, 224, device=dev)
for cl in (False, True):
    m = zoo.build_classifier("resnet50", seed=0, device=dev, channels_last=cl)
    lab = engine.predict(m, x)
    for i in range(6):
        if i == 2:
            torch.cuda.synchronize(); t0 = time.perf_counter()
        engine.input_gradient(m, x, lab, "logits", -1.0, 50.0, "sum")
    torch.cuda.synchronize()
    print(f"channels_last={cl}: {(time.perf_counter() - t0) / 4 * 1e3:.1f} ms per fwd+bwd", flush=True)
