"""Dataset views feeding the ADiL path — drop-in for the reference's imagenet_loading.py.

`Subset_I` carries the `indexed` protocol the learner depends on (adil.py:117,129,168): with indexed=True an
item is (index, x, y), otherwise (x, y).  `dataset_split_by_class` assumes, like upstream, the ImageNet
validation layout of 50 images per class.  `SyntheticImageNet` is an addition: a seeded in-memory stand-in of
the right shape for boxes without the ILSVRC files (no network here)."""
import random

import numpy as np
import torch
from torch.utils.data import Dataset, Subset


class Subset_I(Subset):
    """imagenet_loading.py:8-18."""

    def __init__(self, dataset, indices, indexed=False):
        super().__init__(dataset=dataset, indices=indices)
        self.indexed = indexed

    def __getitem__(self, item):
        x, y = super().__getitem__(item)
        return (item, x, y) if self.indexed else (x, y)

    def __getitems__(self, items):
        # torch >= 2.1 DataLoaders fetch a Subset through __getitems__, which would bypass the indexed protocol
        return [self[i] for i in items]


def dataset_split_by_class(dataset, number_per_class, number_of_classes=1000, samples_per_class=50):
    """Class-balanced train/val/test split (imagenet_loading.py:21-44): per class, shuffle its images with the
    `random` module's global RNG and take the first n_train, next n_val, next n_test."""
    labels = [lab for (_, lab) in dataset.samples]
    order = np.argsort(labels, kind="stable")
    per_class = order.reshape(len(dataset.classes), samples_per_class)
    for row in per_class:
        random.shuffle(row)
    n_tr, n_va, n_te = number_per_class
    train = per_class[:number_of_classes, :n_tr].flatten()
    val = per_class[:number_of_classes, n_tr:n_tr + n_va].flatten()
    test = per_class[:number_of_classes, n_tr + n_va:n_tr + n_va + n_te].flatten()
    return Subset_I(dataset, train), Subset_I(dataset, val), Subset_I(dataset, test)


def load_ImageNet(imagenet_file='./data/ImageNet/ImageNet1000_unnormalized.bin'):
    """imagenet_loading.py:47-56: the pickled dataset object and its class names."""
    dataset = torch.load(imagenet_file, weights_only=False)
    return dataset, dataset.classes


class SyntheticImageNet(Dataset):
    """Seeded synthetic images (3,size,size) in [0,1] with the `.samples` / `.classes` attributes the split relies on.

    structured=False: U[0,1) noise with arbitrary labels — the shape of the data, nothing to classify (a classifier is
    right on ~1/num_classes of it, so performance.py's correctly-classified filter keeps next to nothing).
    structured=True: class c = a coarse `cells` x `cells` colour pattern (fixed by `seed`), bilinearly upsampled, plus
    per-pixel Gaussian noise — images a classifier CAN separate (zoo.fit_centroid_head fits the last layer of a
    random-weight network to them in closed form), so that the whole pipeline — learn, attack, evaluate — reports real
    fooling rates on a box without the ILSVRC files."""

    def __init__(self, num_classes=10, samples_per_class=50, size=224, seed=0, structured=False, noise=0.10, cells=7):
        self.classes = [f"class_{i}" for i in range(num_classes)]
        self.samples = [(f"synthetic_{i}", i // samples_per_class) for i in range(num_classes * samples_per_class)]
        self.size, self.seed, self.structured, self.noise = size, seed, bool(structured), float(noise)
        self.prototypes = None
        if self.structured:
            g = torch.Generator().manual_seed(seed)
            protos = torch.rand(num_classes, 3, cells, cells, generator=g)
            self.prototypes = torch.nn.functional.interpolate(protos, size=(size, size), mode="bilinear",
                                                              align_corners=False) * 0.6 + 0.2

    def __len__(self):
        return len(self.samples)

    def __getitem__(self, item):
        g = torch.Generator().manual_seed(self.seed * 1_000_003 + int(item))
        label = self.samples[item][1]
        if not self.structured:
            return torch.rand(3, self.size, self.size, generator=g), label
        x = self.prototypes[label] + self.noise * torch.randn(3, self.size, self.size, generator=g)
        return x.clamp_(0.0, 1.0), label
