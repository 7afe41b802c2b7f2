"""The data step in front of the path: DS_ImageNet.py — folder scan, Resize(256) / CenterCrop(224) /
ToTensor with torchvision's arithmetic, the pickled `.bin` that imagenet_loading.load_ImageNet reads, the class-balanced split."""
import os
import random

import numpy as np
import pytest
import torch

PIL = pytest.importorskip("PIL")
from PIL import Image

import DS_ImageNet as D
import imagenet_loading as L


def test_resize_and_crop_arithmetic_is_torchvisions():
    # Resize(256): shorter side 256, longer side TRUNCATED — int(256 * 300 / 257) = 298, not round() = 299
    assert D.resized_size(500, 375) == (341, 256)
    assert D.resized_size(375, 500) == (256, 341)
    assert D.resized_size(300, 257) == (298, 256)
    assert D.resized_size(256, 256) == (256, 256)
    # CenterCrop(224): offsets are round()ed half to even — 58.5 -> 58 but 59.5 -> 60 (floor would give 59)
    assert D.center_crop_box(341, 256) == (58, 16, 282, 240)
    assert D.center_crop_box(343, 256) == (60, 16, 284, 240)
    assert D.center_crop_box(256, 256) == (16, 16, 240, 240)


def _tree(root, spec, fmt="JPEG"):
    rng = np.random.RandomState(0)
    os.makedirs(os.path.join(root, "ILSVRC"), exist_ok=True)
    with open(os.path.join(root, D.LABLE_PATH), "w") as f:
        for wnid, names, _ in spec:
            f.write(f"{wnid} {names}\n")
    for wnid, _, files in reversed(spec):                     # written out of order: the scan must sort
        d = os.path.join(root, D.VALID_PATH, wnid)
        os.makedirs(d)
        for name, (w, h) in reversed(files):
            Image.fromarray(rng.randint(0, 256, (h, w, 3), dtype=np.uint8)).save(os.path.join(d, name), fmt)
        open(os.path.join(d, "notes.txt"), "w").write("not an image")


SPEC = [("n01440764", "tench, Tinca tinca", [("ILSVRC2012_val_00000293.JPEG", (500, 375)), ("ILSVRC2012_val_00002138.JPEG", (300, 257))]),
        ("n01443537", "goldfish, Carassius auratus", [("ILSVRC2012_val_00000236.JPEG", (375, 500)), ("ILSVRC2012_val_00000262.JPEG", (343, 256))]),
        ("n01484850", "great white shark, white shark", [("ILSVRC2012_val_00002338.JPEG", (256, 256)), ("ILSVRC2012_val_00002752.JPEG", (640, 480))])]


def test_folder_dataset_items_and_attributes(tmp_path):
    _tree(str(tmp_path), SPEC)
    ds = D.DS_ImageNet(str(tmp_path), split="val", transform=D.transform)
    assert len(ds) == 6 and ds.targets == [0, 0, 1, 1, 2, 2]
    assert [os.path.basename(p) for p, _ in ds.samples] == [f for _, _, fs in SPEC for f, _ in fs]
    assert ds.classes == ["tench", "goldfish", "great white shark"]          # first name of the synset (DS_ImageNet.py:42)
    assert ds.class_to_idx == {"n01440764": 0, "n01443537": 1, "n01484850": 2}
    assert ds.idx_to_class[2] == "n01484850" and ds.classes_to_wnids["n01443537"].startswith("goldfish")
    for i in range(len(ds)):
        x, y = ds[i]
        assert x.shape == (3, 224, 224) and x.dtype == torch.float32 and y == ds.targets[i]
        assert 0.0 <= float(x.min()) and float(x.max()) <= 1.0
        im = Image.open(ds.samples[i][0]).convert("RGB")                   # the three transforms by hand
        im = im.resize(D.resized_size(*im.size), Image.BILINEAR)
        im = im.crop(D.center_crop_box(*im.size))
        ref = torch.from_numpy(np.array(im)).permute(2, 0, 1).float() / 255
        assert torch.equal(x, ref)
    raw = D.DS_ImageNet(str(tmp_path), split="val")                         # no transform: the PIL image itself
    assert raw[1][0].size == (300, 257)
    with pytest.raises(FileNotFoundError):
        D.DS_ImageNet(str(tmp_path), split="train")


def test_saved_dataset_feeds_load_imagenet_and_the_split(tmp_path):
    spec = [(w, n, [(f"img_{k}.png", (260 + k, 256)) for k in range(4)]) for w, n, _ in SPEC]
    _tree(str(tmp_path), spec, fmt="PNG")

    class A:
        root, split, save, file_samples_dataset = str(tmp_path), "val", True, "ImageNet1000_unnormalized.bin"
    D.main(A)
    dataset, classes = L.load_ImageNet(os.path.join(str(tmp_path), A.file_samples_dataset))
    assert classes == ["tench", "goldfish", "great white shark"] and len(dataset) == 12
    random.seed(3)
    train, val, test = L.dataset_split_by_class(dataset, (2, 1, 1), number_of_classes=3, samples_per_class=4)
    assert (len(train), len(val), len(test)) == (6, 3, 3)
    seen = sorted(list(train.indices) + list(val.indices) + list(test.indices))
    assert seen == list(range(12))
    train.indexed = True
    item, x, y = train[0]
    assert item == 0 and x.shape == (3, 224, 224) and y == dataset.targets[train.indices[0]]


def test_main_py_load_image_is_the_same_transform(tmp_path):
    import main as M
    p = os.path.join(str(tmp_path), "a.png")
    Image.fromarray(np.random.RandomState(1).randint(0, 256, (257, 300, 3), dtype=np.uint8)).save(p)
    assert torch.equal(M.load_image(p), D.transform(Image.open(p)))


class _Labelled(torch.utils.data.Dataset):
    def __init__(self, x, y):
        self.x, self.y = x, y

    def __len__(self):
        return len(self.y)

    def __getitem__(self, i):
        return self.x[i], int(self.y[i])


def _accuracy_problem(n=300):
    g = torch.Generator().manual_seed(0)
    x = torch.randn(n, 3, 4, 4, generator=g)
    net = torch.nn.Sequential(torch.nn.Flatten(), torch.nn.Linear(48, 5))
    with torch.no_grad():
        truth = net(x).argmax(-1)
    y = truth.clone()
    y[::4] = (y[::4] + 1) % 5                                  # every fourth label is wrong: accuracy = 225 / 300
    return _Labelled(x, y), net


def test_model_accuracy_is_correct_over_seen():
    from model_accuracy import model_accuracy
    data, net = _accuracy_problem()
    acc = model_accuracy(data, net, device="cpu")
    assert acc.dim() == 0 and acc.dtype == torch.float32 and abs(float(acc) - 0.75) < 1e-7
    assert abs(float(model_accuracy(data, net.to(torch.bfloat16), device="cpu", batch_size=7)) - 0.75) < 0.03   # inputs follow the model's dtype
    assert float(model_accuracy(_Labelled(data.x[:0], data.y[:0]), net, device="cpu")) == 0.0


def _accuracy_worker(rank, world, port, out):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.set_num_threads(1)
    from dl_attack_on_imagenet_amd import dist as adist
    from model_accuracy import model_accuracy_distributed
    adist.init_from_env(backend="gloo")
    data, net = _accuracy_problem(301)                          # 301 images: the shards are uneven, nothing is counted twice
    torch.save(model_accuracy_distributed(data, net, "cpu", batch_size=16), os.path.join(out, f"acc{rank}.pt"))
    torch.distributed.barrier()
    torch.distributed.destroy_process_group()


@pytest.mark.timeout(300)
def test_model_accuracy_distributed_counts_every_image_once(tmp_path):
    import socket
    import torch.multiprocessing as mp
    from model_accuracy import model_accuracy
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    mp.start_processes(_accuracy_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True, start_method="spawn")
    data, net = _accuracy_problem(301)
    want = float(model_accuracy(data, net))
    got = [float(torch.load(os.path.join(str(tmp_path), f"acc{r}.pt"))) for r in range(2)]
    assert got[0] == got[1] and abs(got[0] - want) < 1e-7
