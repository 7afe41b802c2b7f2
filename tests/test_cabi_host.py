"""CPU: the C-ABI library loads and exports every symbol include/adil_hip.h declares (no compute calls without a
GPU), the ctypes table mirrors the header, the product refuses to run on CPU tensors, and the host-side logic
(AdamW scalars, sharding, hyper-parameter grid, CLI flags, model zoo, dataset protocol) behaves."""
import ctypes
import math
import os
import re

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_symbols():
    src = open(os.path.join(ROOT, "include", "adil_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(adil_[a-z0-9_]+)\s*\(", src)))


def test_library_exports_every_declared_symbol():
    from dl_attack_on_imagenet_amd import _lib
    from dl_attack_on_imagenet_amd.build import build_library
    build_library(verbose=False)
    lib = ctypes.CDLL(_lib.LIBPATH)
    syms = header_symbols()
    assert len(syms) >= 18
    for name in syms:
        assert hasattr(lib, name), f"{name} declared in adil_hip.h but not exported"
    assert sorted(_lib.SIGNATURES) == syms            # binding table and header agree
    bound = _lib.load()
    assert bound.adil_abi_version() == _lib.ABI_VERSION
    assert bound.adil_max_atoms() == 128
    assert bound.adil_grad_workspace_bytes(512, 150528, 50) > 0


def test_missing_library_fails_loudly(tmp_path):
    from dl_attack_on_imagenet_amd import _lib
    with pytest.raises(_lib.AdilLibraryError):
        _lib.load(str(tmp_path / "nope.so"))


def test_product_has_no_cpu_fallback():
    from dl_attack_on_imagenet_amd import ops
    from attacks.utils import project_onto_l1_ball
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        ops.l1ball_project_(torch.zeros(3, 4), 0.1)
    with pytest.raises(RuntimeError):
        project_onto_l1_ball(torch.zeros(3, 4), 0.1)
    with pytest.raises(RuntimeError):
        ops.pack_codes(torch.zeros(3, 4), None, 3)


def test_product_never_imports_the_oracle():
    for base in ("dl_attack_on_imagenet_amd", "attacks"):
        for dirpath, _, files in os.walk(os.path.join(ROOT, base)):
            for f in files:
                if f.endswith(".py"):
                    assert "oracle" not in open(os.path.join(dirpath, f)).read(), os.path.join(dirpath, f)
    for f in ("performance.py", "demo_dL_attack.py", "main.py", "imagenet_loading.py", "DS_ImageNet.py", "model_accuracy.py"):
        assert "oracle" not in open(os.path.join(ROOT, f)).read(), f
    for f in os.listdir(os.path.join(ROOT, "tools")):                      # developer tools measure the product, never the oracle
        if f.endswith((".py", ".sh")):
            assert "oracle" not in open(os.path.join(ROOT, "tools", f)).read(), f


def test_adamw_schedule_matches_torch():
    from dl_attack_on_imagenet_amd.ops import AdamWSchedule
    from oracle.adil_oracle import AdamWState
    g = torch.Generator().manual_seed(0)
    p0 = torch.randn(257, generator=g)
    ref = torch.nn.Parameter(p0.clone())
    opt = torch.optim.AdamW([ref], lr=0.01)
    sch = AdamWSchedule(0.01)
    p = p0.clone()
    st = AdamWState(p, 0.01)
    m, s = torch.zeros_like(p0), torch.zeros_like(p0)
    q = p0.clone()
    for _ in range(4):
        grad = torch.randn(257, generator=g)
        ref.grad = grad.clone()
        opt.step()
        st.step(p, grad)
        h = sch.next()                               # the scalars handed to the kernel, applied in plain torch
        q = q * h.decay
        m = m + (1 - h.b1) * (grad - m)
        s = s * h.b2 + (1 - h.b2) * grad * grad
        q = q - h.step_size * (m / (s.sqrt() / h.bc2_sqrt + h.eps))
        assert float((ref.data - p).abs().max()) < 1e-6
        assert float((ref.data - q).abs().max()) < 1e-6
    assert sch.t == 4 and math.isclose(h.step_size, 0.01 / (1 - 0.9 ** 4))


def test_shard_bounds_cover_and_balance():
    from dl_attack_on_imagenet_amd.dist import shard_batch, shard_bounds
    for n in (1, 7, 512, 4096, 10001):
        for w in (1, 2, 3, 8):
            spans = [shard_bounds(n, r, w) for r in range(w)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(spans[i][1] == spans[i + 1][0] for i in range(w - 1))
            sizes = [b - a for a, b in spans]
            assert max(sizes) - min(sizes) <= 1
    assert shard_batch(list(range(10)), 1, 4) == [3, 4, 5]


def test_get_args_grid_and_slices():
    import performance as perf
    from attacks.utils import get_slices
    assert perf.get_args(()) == [{}]
    assert perf.get_args(("a", [1, 2])) == [{"a": 1}, {"a": 2}]
    grid = perf.get_args(("a", [1, 2], "b", [3], "c", [4, 5]))
    assert grid == [{"a": 1, "b": 3, "c": 4}, {"a": 1, "b": 3, "c": 5}, {"a": 2, "b": 3, "c": 4}, {"a": 2, "b": 3, "c": 5}]
    assert get_slices(7, 3) == [[0, 1, 2], [3, 4, 5], [6]]
    assert get_slices(0, 3) == []


def test_cli_flags_match_reference():
    import demo_dL_attack
    import main as main_cli
    a = demo_dL_attack.build_parser().parse_args([])
    assert (a.model, a.seed, a.num_train_per_class, a.trained_classes, a.distributed, a.gpu, a.steps_inference) == \
        ("mobilenet", 3, 1, 1000, False, 0, 100)
    a = demo_dL_attack.build_parser().parse_args(["-m", "densenet", "-s", "5", "--steps-inference", "30"])
    assert (a.model, a.seed, a.steps_inference) == ("densenet", 5, 30)
    assert main_cli.build_parser().parse_args([]).model == "mobilenet"
    assert demo_dL_attack.main(a) is None            # no GPU here: returns early like the reference


def test_zoo_names_and_shapes():
    from dl_attack_on_imagenet_amd import zoo
    assert zoo.canonical_name("resnet") == "resnet18" and zoo.canonical_name("DenseNet") == "densenet121"
    assert zoo.canonical_name("mobilenet") == "mobilenet_v2" and zoo.canonical_name("vgg") == "vgg11"
    assert zoo.canonical_name("googlenet") == "googlenet" and zoo.canonical_name("inception") == "inception_v3"
    with pytest.raises(ValueError):
        zoo.canonical_name("alexnet")
    # all six names of the reference CLI (demo_dL_attack.py:41-53) + the BASELINE.json ones; parameter counts are the
    # published ones of the torchvision definitions (googlenet / inception_v3 without their training-only aux heads)
    for name, nparam in (("resnet18", 11689512), ("resnet50", 25557032), ("densenet121", 7978856),
                         ("googlenet", 6624904), ("inception", 23834568), ("mobilenet", 3504872), ("vgg", 132863336)):
        m = zoo.build_classifier(name, seed=1)
        assert sum(p.numel() for p in m[1].parameters()) == nparam          # torchvision-compatible definitions
        assert not any(p.requires_grad for p in m.parameters()) and not m.training
    y = zoo.build_classifier("resnet18", num_classes=7, seed=1)(torch.rand(2, 3, 64, 64))
    assert y.shape == (2, 7)
    a = zoo.build_classifier("resnet18", seed=3)[1].fc.weight
    b = zoo.build_classifier("resnet18", seed=3)[1].fc.weight
    assert torch.equal(a, b)


def test_weights_round_trip_with_torchvision_key_names(tmp_path):
    """`--weights`: a state_dict with torchvision's key names on local disk loads into the zoo definitions (the keys
    listed are the well-known ones of torchvision's checkpoints), training-only aux heads in a checkpoint are ignored,
    and the loaded network computes the checkpoint's function, not the seed's."""
    from dl_attack_on_imagenet_amd import zoo
    expected = {
        "resnet18": ["conv1.weight", "bn1.running_var", "layer1.0.conv1.weight", "layer2.0.downsample.0.weight",
                     "layer4.1.bn2.num_batches_tracked", "fc.bias"],
        "resnet50": ["layer1.0.conv3.weight", "layer1.0.downsample.1.running_mean", "layer4.2.bn3.weight", "fc.weight"],
        "densenet121": ["features.conv0.weight", "features.denseblock1.denselayer1.norm1.weight",
                        "features.denseblock4.denselayer16.conv2.weight", "features.transition3.conv.weight",
                        "features.norm5.bias", "classifier.weight"],
        "vit_b_16": ["class_token", "conv_proj.weight", "encoder.pos_embedding",
                     "encoder.layers.encoder_layer_0.self_attention.in_proj_weight",
                     "encoder.layers.encoder_layer_11.mlp.3.bias", "encoder.ln.weight", "heads.head.weight"],
        "mobilenet_v2": ["features.0.0.weight", "features.1.conv.0.0.weight", "features.18.1.running_var", "classifier.1.weight"],
        "vgg11": ["features.0.weight", "features.18.bias", "classifier.0.weight", "classifier.6.bias"],
        "googlenet": ["conv1.conv.weight", "conv1.bn.running_mean", "inception3a.branch2.1.conv.weight",
                      "inception5b.branch4.1.bn.weight", "fc.weight"],
        "inception_v3": ["Conv2d_1a_3x3.conv.weight", "Mixed_5b.branch5x5_2.conv.weight", "Mixed_6e.branch7x7dbl_5.bn.bias",
                         "Mixed_7c.branch3x3dbl_3b.conv.weight", "fc.bias"],
    }
    for name, keys in expected.items():
        sd = zoo._BUILDERS[name](1000).state_dict()
        missing = [k for k in keys if k not in sd]
        assert not missing, (name, missing)
    for name in ("resnet18", "googlenet"):
        src = zoo.build_classifier(name, seed=11)
        state = dict(src[1].state_dict())
        if name == "googlenet":                                   # torchvision's googlenet checkpoint carries the aux heads
            state["aux1.conv.conv.weight"] = torch.zeros(128, 512, 1, 1)
            state["aux2.fc2.bias"] = torch.zeros(1000)
        path = tmp_path / f"{name}.pth"
        torch.save(state, path)
        loaded = zoo.build_classifier(name, seed=99, weights=str(path))
        other = zoo.build_classifier(name, seed=99)
        x = torch.rand(2, 3, 64, 64)
        assert torch.equal(loaded(x), src(x)) and not torch.equal(other(x), src(x))
    with pytest.raises(RuntimeError):                            # a checkpoint of another architecture fails loudly
        zoo.build_classifier("resnet50", weights=str(tmp_path / "resnet18.pth"))


def test_epilogue_tables_stay_fp32_under_dtype_casts():
    """FusedResNet's BatchNorm scale/shift tables are derived in fp64 and must survive `.to(bfloat16)` unrounded."""
    from dl_attack_on_imagenet_amd import zoo
    net = zoo.ResNet(zoo.Bottleneck, [1, 1, 1, 1])
    g = torch.Generator().manual_seed(0)
    for m in net.modules():
        if isinstance(m, torch.nn.BatchNorm2d):
            m.running_var.copy_(0.5 + torch.rand(m.running_var.shape, generator=g))
            m.running_mean.copy_(torch.randn(m.running_mean.shape, generator=g))
            m.weight.data.copy_(0.5 + torch.rand(m.weight.shape, generator=g))
            m.bias.data.copy_(torch.randn(m.bias.shape, generator=g))
    want = zoo._bn_affine(net.bn1)
    fused = zoo.FusedResNet(net, normalize=([0.485, 0.456, 0.406], [0.229, 0.224, 0.225]))
    fused = fused.to(dtype=torch.bfloat16).to(memory_format=torch.channels_last)
    stem = [m for m in fused.modules() if isinstance(m, zoo._FusedStem)][0]
    assert stem.scale.dtype == torch.float32 and torch.equal(stem.scale, want[0]) and torch.equal(stem.shift, want[1])
    tables = [m for m in fused.modules() if isinstance(m, zoo._Fp32Tables)]
    assert len(tables) > 10 and all(m.scale.dtype == m.shift.dtype == torch.float32 for m in tables)
    assert all(p.dtype == torch.bfloat16 for p in fused.parameters())


def test_shuffled_batches_follow_the_reference_dataloader():
    """loader.shuffled_batches consumes the global torch RNG exactly like `DataLoader(shuffle=True)` (adil.py:130-133):
    same seed -> same index batches, epoch after epoch, with a second (validation) loader interleaved."""
    from dl_attack_on_imagenet_amd.loader import shuffled_batches
    torch.manual_seed(3)
    tr = torch.utils.data.DataLoader(torch.arange(37), batch_size=5, shuffle=True)
    va = torch.utils.data.DataLoader(torch.arange(11), batch_size=5, shuffle=True)
    want = []
    for _ in range(3):
        want.append(([b.tolist() for b in tr], [b.tolist() for b in va]))
    torch.manual_seed(3)
    got = [(shuffled_batches(37, 5), shuffled_batches(11, 5)) for _ in range(3)]
    assert got == want
    assert shuffled_batches(7, 3, shuffle=False) == [[0, 1, 2], [3, 4, 5], [6]]


def test_dataset_protocol_and_split():
    import random
    from imagenet_loading import Subset_I, SyntheticImageNet, dataset_split_by_class
    ds = SyntheticImageNet(num_classes=4, samples_per_class=50, size=8, seed=1)
    random.seed(0)
    tr, va, te = dataset_split_by_class(ds, [3, 2, 5], number_of_classes=4)
    assert (len(tr), len(va), len(te)) == (12, 8, 20)
    assert not (set(tr.indices) & set(va.indices)) and not (set(va.indices) & set(te.indices))
    labels = sorted(ds.samples[i][1] for i in tr.indices)
    assert labels == [0] * 3 + [1] * 3 + [2] * 3 + [3] * 3                   # class balanced
    x, y = tr[0]
    assert x.shape == (3, 8, 8) and 0 <= float(x.min()) and float(x.max()) < 1
    tr.indexed = True
    i, x2, y2 = tr[0]
    assert i == 0 and torch.equal(x, x2) and y == y2
    assert isinstance(tr, Subset_I)
    batch = next(iter(torch.utils.data.DataLoader(tr, batch_size=4, shuffle=False)))      # the path ADIL uses
    assert len(batch) == 3 and batch[0].tolist() == [0, 1, 2, 3] and batch[1].shape == (4, 3, 8, 8)
    tr.indexed = False
    assert len(next(iter(torch.utils.data.DataLoader(tr, batch_size=4)))) == 2


def test_adil_constructor_surface_without_data(tmp_path):
    from attacks import ADIL, ADILR, FastUAP, UAPPGD
    net = torch.nn.Sequential(torch.nn.Flatten(), torch.nn.Linear(12, 3))
    atk = ADIL(net, eps=8 / 255, model_name="none", alpha=0.0, dict_dir=str(tmp_path))      # alpha accepted (Q14)
    assert atk.model_file.endswith("ImageNet_none.bin") and atk.n_atoms == 100 and atk.loss == "ce"
    assert atk.attack == "supervised" and atk.steps_inference == 30 and atk.norm == "linf"
    assert atk.device == torch.device("cpu") and "ADIL" in str(atk)
    with pytest.raises(FileNotFoundError):
        atk(torch.zeros(1, 3, 2, 2), torch.zeros(1, dtype=torch.long))
    for cls in (ADILR, FastUAP):
        with pytest.raises(NotImplementedError):
            cls(net)
    uap = UAPPGD(net, steps=2, batch_size=4, norm='linf', eps=0.1, model_dir=str(tmp_path))   # no data: nothing is learned
    assert uap.model_name.endswith("UAPPGD_model_test.bin") and uap.optimizer == 'adam' and uap.beta == 9
    with pytest.raises(RuntimeError):                     # learning runs on the HIP kernels only: no CPU fallback
        uap.learn_attack(dataset=[(torch.zeros(3, 2, 2), 0)] * 4)
    lg = torch.tensor([[1.0, 3.0, -2.0], [-1.0, -3.0, -2.0]])
    out = atk.f_loss(lg, torch.tensor([1, 0]))
    assert out.tolist() == [2.0, -1.0]               # second row: other logits negative -> max is the zeroed label (Q5)


def test_vit_patch_embedding_is_the_convolution():
    """zoo.VisionTransformer computes conv_proj as an unfold + GEMM (same parameters): identical function, and its input
    gradient equals the convolution's."""
    from dl_attack_on_imagenet_amd import zoo
    torch.manual_seed(0)
    net = zoo.VisionTransformer(image_size=64, layers=1, heads=4, dim=64, mlp_dim=128, num_classes=5)
    x = torch.rand(2, 3, 64, 64, requires_grad=True)
    a = net._patch_embed(x)
    b = net.conv_proj(x).reshape(2, 64, -1).permute(0, 2, 1)
    assert float((a - b).abs().max()) < 1e-5
    w = torch.randn_like(a)
    (ga,) = torch.autograd.grad((a * w).sum(), x)
    (gb,) = torch.autograd.grad((b * w).sum(), x)
    assert float((ga - gb).abs().max()) < 1e-5


def test_main_load_image_follows_the_reference_transform(tmp_path):
    """main.load_image = Resize(256) / CenterCrop(224) / ToTensor of the reference (main.py:64-75, DS_ImageNet.py:14-18)
    without torchvision: shorter side to 256 (bilinear), central 224 x 224 crop, float CHW in [0, 1]."""
    PIL = pytest.importorskip("PIL.Image")
    import numpy as np
    import main as main_cli
    rng = np.random.default_rng(0)
    arr = rng.integers(0, 256, size=(300, 480, 3), dtype=np.uint8)           # H=300, W=480
    path = tmp_path / "img.png"
    PIL.fromarray(arr).save(path)
    x = main_cli.load_image(str(path))
    assert x.shape == (3, 224, 224) and x.dtype == torch.float32 and 0.0 <= float(x.min()) and float(x.max()) <= 1.0
    # torchvision's arithmetic: the longer side is TRUNCATED (int(409.6) = 409), the crop offset rounds half to even (92.5 -> 92)
    im = PIL.open(path).convert("RGB").resize((int(480 * 256 / 300), 256), PIL.BILINEAR)     # (409, 256)
    left, top = int(round((im.size[0] - 224) / 2.0)), (256 - 224) // 2
    want = torch.from_numpy(np.asarray(im.crop((left, top, left + 224, top + 224)), dtype=np.float32) / 255).permute(2, 0, 1)
    assert torch.equal(x, want)
    # a portrait image: the WIDTH becomes 256
    PIL.fromarray(arr.transpose(1, 0, 2)).save(path)
    assert main_cli.load_image(str(path)).shape == (3, 224, 224)


def test_margin_loss_matches_the_reference_f_loss_fixture():
    """The PRODUCT's f_loss (engine.margin_loss, ADIL.f_loss -> adil.py:103-112) against fixture G5, which holds the
    reference's own values and gradients on random logits incl. the row whose other logits are all negative (the zeroed
    label logit then wins the max — quirk Q5): value and gradient exact."""
    from conftest import load_golden, t
    from dl_attack_on_imagenet_amd import engine
    z = load_golden("g5_floss")
    lg = t(z["logits"]).clone().requires_grad_(True)
    val = engine.margin_loss(lg, t(z["labels"]), float(z["kappa"]))
    val.sum().backward()
    assert torch.equal(val.detach(), t(z["value"]))
    assert torch.equal(lg.grad, t(z["grad"]))
    # attack_loss('logits') is its sum (adil.py:183), whatever the logits' dtype
    tot = engine.attack_loss(t(z["logits"]).to(torch.bfloat16), t(z["labels"]), "logits", -1.0, float(z["kappa"]), "sum")
    assert tot.dtype == torch.float32 and abs(float(tot) - float(t(z["value"]).sum())) < 0.5


def test_demo_cast_loader_keeps_the_loader_interface():
    """the bf16 demo path used to turn its DataLoader into a list, which lost `batch_size` (the
    classifier's batch bucket) and `dataset` (the transfer evaluation's sample count)."""
    import demo_dL_attack as demo
    import performance as perf
    ds = torch.utils.data.TensorDataset(torch.rand(7, 3, 4, 4), torch.arange(7))
    loader = torch.utils.data.DataLoader(ds, batch_size=3, shuffle=False)
    assert demo._cast_loader(loader, torch.float32) is loader
    cast = demo._cast_loader(loader, torch.bfloat16)
    assert perf._loader_batch_size(cast) == 3 and len(cast.dataset) == 7 and len(cast) == 3
    for _ in range(2):                                             # re-iterable, like a DataLoader
        got = list(cast)
        assert [x.shape[0] for x, _ in got] == [3, 3, 1] and all(x.dtype == torch.bfloat16 for x, _ in got)
        assert torch.equal(torch.cat([y for _, y in got]), torch.arange(7))


def test_bench_defaults_by_mode_and_traffic_is_tied_to_the_build(tmp_path, monkeypatch):
    """bench.py: (i) the per-mode defaults (learn / inference: configs[1]'s 50 atoms, 100 steps; transfer: the reference's
    100 atoms, 4 batches, 100 DDrague iterations; cached pseudo-labels by default); (ii) `roofline.traffic` is reported only
    when profiles/hbm_traffic*.json was measured on THIS build of the kernels: a file recorded for another
    source hash, or for another atom count, gives None."""
    import json
    import bench
    from dl_attack_on_imagenet_amd.build import source_hash
    monkeypatch.setattr("sys.argv", ["bench.py"])
    a = bench.parse_args()
    assert (a.mode, a.steps, a.warmup, a.atoms, a.cache_labels, a.gpus) == ("learn", 100, 5, 50, 1, 1)
    monkeypatch.setattr("sys.argv", ["bench.py", "--mode", "transfer"])
    a = bench.parse_args()
    assert (a.steps, a.warmup, a.atoms, a.steps_inference) == (4, 1, 100, 100)
    monkeypatch.setattr("sys.argv", ["bench.py", "--mode", "transfer", "--steps", "2", "--atoms", "10"])
    a = bench.parse_args()
    assert (a.steps, a.atoms) == (2, 10)
    os.makedirs(tmp_path / "profiles")
    monkeypatch.setattr(bench, "ROOT", str(tmp_path))
    assert bench.measured_traffic("synth", 50) is None                                   # no file
    rec = {"_source": {"kernel_source_hash": source_hash(), "atoms": 50}, "synth": 341.0e6}
    json.dump(rec, open(tmp_path / "profiles" / "hbm_traffic.json", "w"))
    assert bench.measured_traffic("synth", 50) == 341.0e6
    assert bench.measured_traffic("zstep_", 50) is None                                  # group not measured
    assert bench.measured_traffic("synth", 100) is None                                  # other atom count: its own file
    rec["_source"]["kernel_source_hash"] = "0" * 16                                      # measured on another build
    json.dump(rec, open(tmp_path / "profiles" / "hbm_traffic.json", "w"))
    assert bench.measured_traffic("synth", 50) is None
    assert bench.measured_mfma_utilisation("synth", 50) is None                          # same rule for the MFMA pass: no file
    mrec = {"_source": {"kernel_source_hash": source_hash(), "atoms": 50}, "synth": 0.05}
    json.dump(mrec, open(tmp_path / "profiles" / "mfma_utilisation.json", "w"))
    assert bench.measured_mfma_utilisation("synth", 50) == 0.05 and bench.measured_mfma_utilisation("synth", 100) is None
    mrec["_source"]["kernel_source_hash"] = "0" * 16
    json.dump(mrec, open(tmp_path / "profiles" / "mfma_utilisation.json", "w"))
    assert bench.measured_mfma_utilisation("synth", 50) is None
    # the committed files should belong to the committed kernel tree (else the bench line carries nulls for them); a warning,
    # not a failure: while a kernel is being changed the counters are stale until `tools/run_profiles.sh` has run again
    import warnings
    for name in ("hbm_traffic.json", "hbm_traffic_k100.json", "mfma_utilisation.json", "mfma_utilisation_k100.json"):
        committed = json.load(open(os.path.join(ROOT, "profiles", name)))
        if committed["_source"]["kernel_source_hash"] != source_hash():
            warnings.warn(f"profiles/{name} was measured on another kernel tree: re-run tools/run_profiles.sh pmc / mfma")
    a50 = bench.algorithmic_bytes(512, 150528, 50, 512, 2, "learn")
    assert a50["synth"] == 2 * 512 * 150528 * 2 + 150528 * 50 * 4 + 512 * 50 * 4          # 338.5 MB


def test_structured_synthetic_dataset_is_seeded_and_separable():
    """imagenet_loading.SyntheticImageNet(structured=True): deterministic items, images of one class closer to each other than
    to another class, the `.samples` / `.classes` attributes of the split; structured=False stays the U[0,1) stand-in."""
    from imagenet_loading import SyntheticImageNet
    ds = SyntheticImageNet(num_classes=3, samples_per_class=50, size=32, seed=2, structured=True)
    x0, y0 = ds[0]
    x0b, _ = ds[0]
    x1, y1 = ds[1]
    x50, y50 = ds[50]
    assert torch.equal(x0, x0b) and (y0, y1, y50) == (0, 0, 1) and x0.shape == (3, 32, 32)
    assert float(x0.min()) >= 0.0 and float(x0.max()) <= 1.0
    assert float((x0 - x1).abs().mean()) < float((x0 - x50).abs().mean())
    flat = SyntheticImageNet(num_classes=3, samples_per_class=50, size=32, seed=2)
    assert flat.prototypes is None and not torch.equal(flat[0][0], x0)


def test_fp32_head_keeps_its_bits_and_the_switch_restores():
    """zoo._Fp32Head (round 4): the fp32 copy of the last linear layer survives `.to(bfloat16)` bit for bit and yields fp32
    logits; engine.precise_head switches it on for networks that have it (FusedResNet built with head_fp32="inference"),
    restores the previous state on exit — also when nested or left through an exception — and ignores other models."""
    import torch
    from dl_attack_on_imagenet_amd import engine, zoo
    torch.manual_seed(0)
    fc = torch.nn.Linear(16, 5)
    head = zoo._Fp32Head(fc).to(torch.bfloat16)
    assert head.weight.dtype == torch.float32 and torch.equal(head.weight, fc.weight.detach())
    feats = torch.randn(3, 16, 4, 4).to(torch.bfloat16)
    out = head(feats)
    assert out.dtype == torch.float32
    ref = torch.nn.functional.linear(feats.float().mean(dim=(2, 3)), fc.weight, fc.bias)
    assert torch.allclose(out, ref, atol=1e-6)
    net = zoo._BUILDERS["resnet18"](10)
    fused = zoo.FusedResNet(net, head_fp32="inference")
    model = torch.nn.Sequential(fused)
    assert fused.head32 is not None and not fused.head32_on
    with engine.precise_head(model):
        assert fused.head32_on
        with engine.precise_head(model, False):
            assert not fused.head32_on
        assert fused.head32_on
    assert not fused.head32_on
    try:
        with engine.precise_head(model):
            raise RuntimeError("x")
    except RuntimeError:
        pass
    assert not fused.head32_on
    always = zoo.FusedResNet(zoo._BUILDERS["resnet18"](10), head_fp32=True)
    assert always.head32_on
    plain = zoo.FusedResNet(zoo._BUILDERS["resnet18"](10))
    assert plain.head32 is None and plain.precise_head(True) is False and not plain.head32_on
    with engine.precise_head(torch.nn.Linear(2, 2)):
        pass
    import pytest
    with pytest.raises(ValueError):
        zoo.FusedResNet(zoo._BUILDERS["resnet18"](10), head_fp32="sometimes")
