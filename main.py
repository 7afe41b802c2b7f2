"""One-image demo CLI — drop-in for the reference's main.py: load a classifier, attack one image with a learned
dictionary (`trained_dicts/ImageNet_{model}.bin`), report the label change.  `--model/-m` as upstream
(main.py:109-115); --image/--weights/--synthetic are additions.  The matplotlib figure of the reference is
cosmetic and is only drawn when matplotlib and PIL are importable."""
import argparse

import torch

from attacks import ADIL
from dl_attack_on_imagenet_amd import zoo

DEFAULT_IMAGE = 'data/ImageNet/ILSVRC/Data/val/n01484850/ILSVRC2012_val_00002752.JPEG'   # main.py:69


def load_image(path, size=224):
    """Resize(256) / CenterCrop(224) / ToTensor of the reference (main.py:64-75) without torchvision."""
    from PIL import Image
    from DS_ImageNet import transform
    with open(path, 'rb') as f:
        return transform(Image.open(f), crop=size)


def build_parser():
    p = argparse.ArgumentParser()
    p.add_argument('--model', '-m', metavar='M', default='mobilenet')
    p.add_argument('--image', default=DEFAULT_IMAGE)
    p.add_argument('--weights', default=None)
    p.add_argument('--synthetic', action='store_true', help='attack a seeded random image instead of a JPEG')
    p.add_argument('--image-size', type=int, default=224, help='side of the synthetic image (must match the dictionary)')
    p.add_argument('--figure', default='attack_samples.png')
    p.add_argument('--graph', type=int, default=1,
                   help='replay the inference iterations from a hipGraph (a one-image attack is launch-bound); 0 = eager')
    return p


def main(args):
    if not torch.cuda.is_available():
        print('Check cuda setting for model training on ImageNet')       # main.py:29-31
        return
    torch.cuda.set_device(0)
    device = torch.device('cuda', 0)
    model_name = args.model.lower()          # names the dictionary file, as upstream (main.py:40, adil.py:89-91)
    model = zoo.build_classifier(model_name, weights=args.weights, device=device)
    if args.synthetic:
        im = torch.rand(3, args.image_size, args.image_size, generator=torch.Generator().manual_seed(0))
    else:
        im = load_image(args.image)
    eps = 8 / 255
    attack = ADIL(model, eps=eps, model_name=model_name, use_graph=bool(args.graph))     # main.py:80
    im = im.to(device)
    label = model(im.unsqueeze(0)).argmax(dim=-1)
    adversary = attack(im.unsqueeze(0), label)
    attack_label = model(adversary).argmax(dim=-1)
    print(f'clean label {int(label)} -> adversarial label {int(attack_label)}; '
          f'max|adv-x| = {float((adversary[0] - im).abs().max()):.4f}')
    try:
        import matplotlib
        matplotlib.use('Agg')
        import matplotlib.pyplot as plt
        fig, axes = plt.subplots(1, 3, figsize=(15, 5))
        pert = adversary[0] - im + eps
        for ax, img, title in zip(axes, (im, pert / pert.max(), adversary[0]),
                                  (f'original: {int(label)}', 'perturbation', f'attack: {int(attack_label)}')):
            ax.imshow(img.detach().float().cpu().numpy().transpose(1, 2, 0))
            ax.set_title(title, fontsize=24)
            ax.set_axis_off()
        fig.tight_layout(pad=0.5)
        plt.savefig(args.figure)
    except ImportError:
        pass
    return int(label), int(attack_label)


if __name__ == '__main__':
    main(build_parser().parse_args())
