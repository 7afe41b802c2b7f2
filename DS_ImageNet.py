"""The ILSVRC folder dataset in front of the ADiL path — drop-in for the reference's DS_ImageNet.py, without torchvision.

The reference subclasses torchvision's ImageFolder and composes Resize(256) / CenterCrop(224) / ToTensor
(DS_ImageNet.py:14-18, 34-50); torchvision is not part of this image, so the folder scan and the three transforms are
restated here on PIL + numpy with torchvision's own arithmetic (the output size of Resize truncates, the crop offset
rounds half to even).  What callers rely on is kept: `DS_ImageNet(root, split, transform)`, `len`, `ds[i] -> (x, y)`,
`.classes` (human-readable names, :42), `.class_to_idx`, `.idx_to_class`, `.classes_to_wnids`, `.samples`, `.targets`
(imagenet_loading.dataset_split_by_class and load_ImageNet use `.samples` / `.classes`).

A `.bin` pickled by the REFERENCE (`torch.save(valid_data, ...)`, :57) embeds torchvision objects (ImageFolder's loader,
the Compose) and cannot be unpickled without torchvision; `python DS_ImageNet.py --save` writes the same kind of file from
this class, which `imagenet_loading.load_ImageNet` reads."""
import argparse
import os

import numpy as np
import torch
from torch.utils.data import Dataset

ROOT_PATH = 'ILSVRC'
TRAIN_PATH = os.path.join(ROOT_PATH, 'Data/train')
VALID_PATH = os.path.join(ROOT_PATH, 'Data/val')
LABLE_PATH = os.path.join(ROOT_PATH, 'LOC_synset_mapping.txt')          # (sic) the reference's spelling, DS_ImageNet.py:11
IMG_EXTENSIONS = ('.jpg', '.jpeg', '.png', '.ppm', '.bmp', '.pgm', '.tif', '.tiff', '.webp')   # ImageFolder's list


def resized_size(width, height, size=256):
    """Resize(size) with an int: the shorter side becomes `size`, the longer one int(size * long / short) (truncated)."""
    if width <= height:
        return size, int(size * height / width)
    return int(size * width / height), size


def center_crop_box(width, height, size=224):
    """CenterCrop(size): torchvision rounds the offsets with Python's round (half to even), it does not floor them."""
    left, top = int(round((width - size) / 2.0)), int(round((height - size) / 2.0))
    return left, top, left + size, top + size


def transform(img, resize=256, crop=224):
    """PIL image -> float tensor (3, crop, crop) in [0, 1]: Resize(256) bilinear, CenterCrop(224), ToTensor."""
    from PIL import Image
    img = img.convert('RGB')
    img = img.resize(resized_size(*img.size, resize), Image.BILINEAR)
    img = img.crop(center_crop_box(*img.size, crop))
    return torch.from_numpy(np.asarray(img, dtype=np.uint8).copy()).permute(2, 0, 1).contiguous().float().div_(255.0)


def read_label(path):
    """LOC_synset_mapping.txt: 'n01440764 tench, Tinca tinca' per line -> {wnid: description} (DS_ImageNet.py:21-31)."""
    table = {}
    with open(path, 'r') as f:
        for line in f:
            wnid, _, names = line.rstrip().partition(' ')
            if wnid:
                table[wnid] = names
    return table


class DS_ImageNet(Dataset):
    """`filepath`/ILSVRC/Data/{train,val}/<wnid>/*.JPEG; class index = rank of the wnid folder name, samples in the order
    ImageFolder lists them (classes sorted, then a sorted walk of each class folder)."""

    def __init__(self, filepath, split='train', transform=None, target_transform=None):
        self.split = split
        self.root = os.path.join(filepath, TRAIN_PATH if split == 'train' else VALID_PATH)
        self.transform, self.target_transform = transform, target_transform
        wnids = sorted(e.name for e in os.scandir(self.root) if e.is_dir())
        if not wnids:
            raise FileNotFoundError(f"Couldn't find any class folder in {self.root}.")
        self.class_to_idx = {w: i for i, w in enumerate(wnids)}
        self.samples = []
        for w in wnids:
            found = 0
            for folder, _, names in sorted(os.walk(os.path.join(self.root, w), followlinks=True)):
                for name in sorted(names):
                    if name.lower().endswith(IMG_EXTENSIONS):
                        self.samples.append((os.path.join(folder, name), self.class_to_idx[w]))
                        found += 1
            if not found:
                raise FileNotFoundError(f"Found no valid file for the class {w}.")
        self.imgs = self.samples
        self.targets = [y for _, y in self.samples]
        self.idx_to_class = {i: w for w, i in self.class_to_idx.items()}
        self.classes_to_wnids = read_label(os.path.join(filepath, LABLE_PATH))
        self.classes = [self.classes_to_wnids[w].split(',', 1)[0] for w in wnids]

    def __len__(self):
        return len(self.samples)

    def __getitem__(self, index):
        from PIL import Image
        path, target = self.samples[index]
        with open(path, 'rb') as f:
            sample = Image.open(f).convert('RGB')
        if self.transform is not None:
            sample = self.transform(sample)
        if self.target_transform is not None:
            target = self.target_transform(target)
        return sample, target


def main(args):
    data = DS_ImageNet(args.root, split=args.split, transform=transform)
    print(f'{len(data)} images, {len(data.classes)} classes under {data.root}')
    if args.save:                                   # the reference keeps this line commented out (DS_ImageNet.py:57)
        torch.save(data, os.path.join(args.root, args.file_samples_dataset))
    return data


if __name__ == '__main__':
    argparser = argparse.ArgumentParser('ImageNet management')
    argparser.add_argument('--root', '-r', metavar='R', default='./data/ImageNet',
                           help='ImageNet root file path from current project (default "./data/ImageNet")')
    argparser.add_argument('--split', metavar='S', default='val', help='train or val (default val)')
    argparser.add_argument('--file-samples-dataset', metavar='P', default='ImageNet1000_unnormalized.bin')
    argparser.add_argument('--save', action='store_true', help='pickle the dataset object to ROOT/FILE (addition)')
    main(argparser.parse_args())
