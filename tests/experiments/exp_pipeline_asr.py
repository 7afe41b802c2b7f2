"""Experiment: the reference's ASR pipeline on the structured workload, both legs — learn D on a training split, then
performance() = attack(x, y) (DDrague inference) on a held-out split.  A: fp32 oracle + plain fp32 net; C: the product."""
import json, os, sys, tempfile, time
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE))); sys.path.insert(0, os.path.dirname(HERE))
import torch
import performance as perf
from attacks import ADIL
from dl_attack_on_imagenet_amd import engine, loader, ops
from oracle import adil_oracle as O
from structured import fitted_classifiers, structured_images

E = os.environ.get
n, k, T, S = int(E("N", 512)), int(E("K", 50)), int(E("T", 100)), int(E("S", 100))
eps, dev = 8 / 255, "cuda"
images, labels = structured_images(n, 10, seed=3)
held, held_labels = structured_images(n, 10, seed=3, draw=1)
tmp = tempfile.mkdtemp()
ref, fast, margins, pred = fitted_classifiers("resnet50", images, labels, 10, dev, tmp, target_margin=float(E("MARGIN", 10.0)))
with torch.no_grad():
    acc32 = float(torch.cat([(ref(c.to(dev)).argmax(-1).cpu()) for c in held.split(64)]).eq(held_labels).float().mean())
    acc16 = float(torch.cat([(fast(c.to(dev).to(torch.bfloat16)).argmax(-1).cpu()) for c in held.split(64)]).eq(held_labels).float().mean())
g = torch.Generator().manual_seed(33)
d0 = -1 + 2 * torch.rand(3, 224, 224, k, generator=g)
v0 = O.project_onto_l1_ball(torch.rand(n, k, generator=g), eps)
index = torch.arange(n, device=dev)
out = dict(n=n, k=k, T=T, S=S, heldout_accuracy_fp32=acc32, heldout_accuracy_bf16=acc16, margin_min=float(margins.min()))
# leg A
d, v = d0.clone().to(dev), v0.clone().to(dev)
sd, sv = O.AdamWState(d, 0.01), O.AdamWState(v, 0.01)
x32 = images.to(dev)
t0 = time.time()
fa = [int(O.learn_step_a(ref, x32, index, d, v, sd, sv, eps, "logits", -1.0, 50.0)[1]) for _ in range(T)]
out["learn_A_s"] = time.time() - t0
t0 = time.time()
batches = [(held[lo:lo + 128].to(dev), held_labels[lo:lo + 128].to(dev)) for lo in range(0, n, 128)]
pa = O.performance(lambda xx, yy: O.forward_supervised_ddrague(ref, xx, d, eps, S, "logits"), ref, batches)
out["eval_A_s"] = time.time() - t0
# leg C
x16 = x32.to(torch.bfloat16)
learner = engine.DictionaryLearner(d0.clone().to(dev), v0.clone().to(dev), eps, 0.01, "logits", False, 50.0)
fc = [learner.step(fast, x16, index)[1] for _ in range(T)]
fc = [int(f) for f in fc]
torch.save([learner.d.cpu(), learner.v.cpu(), [], [], torch.tensor(0.)], os.path.join(tmp, "ImageNet_c.bin"))
atk = ADIL(fast, eps=eps, n_atoms=k, attack="supervised", model_name="c", loss="logits", steps_inference=S, dict_dir=tmp,
           stream_dtype=torch.bfloat16)
ds = torch.utils.data.TensorDataset(held, held_labels)
t0 = time.time()
res = loader.ResidentBatches(ds, held_labels, 128, dev, torch.bfloat16)
pc = perf.performance(atk, fast, res)
out["eval_C_s"] = time.time() - t0
# leg P (LEG_P=1): the product in the REFERENCE's configuration — fp32 streams, the fp32 network — end to end
if E("LEG_P", "0") == "1":
    lp = engine.DictionaryLearner(d0.clone().to(dev), v0.clone().to(dev), eps, 0.01, "logits", False, 50.0)
    lab32 = engine.predict(ref, x32)
    fp = [int(lp.step(ref, x32, index, lab32)[1]) for _ in range(T)]
    torch.save([lp.d.cpu(), lp.v.cpu(), [], [], torch.tensor(0.)], os.path.join(tmp, "ImageNet_p.bin"))
    atkp = ADIL(ref, eps=eps, n_atoms=k, attack="supervised", model_name="p", loss="logits", steps_inference=S, dict_dir=tmp)
    out["perf_P_product_fp32_end_to_end"] = {kk: float(vv) for kk, vv in perf.performance(atkp, ref, loader.ResidentBatches(ds, held_labels, 128, dev)).items()}
    out["fooled_learn_P"] = fp[::10] + [fp[-1]]
# cross: the product's dictionary attacked by the oracle inference on the fp32 net, and vice versa
pca = O.performance(lambda xx, yy: O.forward_supervised_ddrague(ref, xx, learner.d, eps, S, "logits"), ref, batches)
out.update(fooled_learn_A=fa[::10] + [fa[-1]], fooled_learn_C=fc[::10] + [fc[-1]], perf_A=pa, perf_C=pc, perf_oracle_inference_with_product_D=pca)
print(json.dumps(out, default=float), flush=True)
