"""Dictionary-learning + evaluation CLI — drop-in for the reference's demo_dL_attack.py on the MI355X engine.

Reference flags are kept (--model/-m, --seed/-s, --num-train-per-class, --trained-classes, --distributed, --gpu,
--steps-inference; demo_dL_attack.py:160-204) with the same defaults and the same hard-coded experiment
(eps 8/255, linf, K=100, 500 steps, lr .01, batch 100, loss 'logits', method 'gd'; :88-118).
Additions, all optional: model names of BASELINE.json (resnet18/resnet50/densenet121/vit_b_16), --weights (local
torchvision state_dict), --synthetic (seeded stand-in dataset when the ILSVRC files are absent), and overrides
--n-atoms/--steps/--batch-size/--dtype for plumbing runs."""
import argparse
import os
import random

import numpy as np
import torch

import performance as perf
from attacks import ADIL
from dl_attack_on_imagenet_amd import zoo
from imagenet_loading import SyntheticImageNet, dataset_split_by_class, load_ImageNet


def build_parser():
    p = argparse.ArgumentParser()
    p.add_argument('--model', '-m', metavar='M', default='mobilenet')
    p.add_argument('--seed', '-s', metavar='S', type=int, default=3, help='change seed to carry out the exp')
    p.add_argument('--num-train-per-class', type=int, default=1, help='number per class for training')
    p.add_argument('--trained-classes', metavar='TC', type=int, default=1000, help='number of class for training')
    p.add_argument('--distributed', metavar='D', type=bool, default=False,
                   help='If distributed data parallel used, default value is False')
    p.add_argument('--gpu', type=int, default=0, help='gpu index, default is 0')
    p.add_argument('--steps-inference', type=int, default=100, help='number of steps for inference, default is 100')
    # additions
    p.add_argument('--weights', default=None, help='local torchvision state_dict for the classifier')
    p.add_argument('--synthetic', action='store_true', help='use a seeded synthetic dataset (no ILSVRC files needed)')
    p.add_argument('--synthetic-classes', type=int, default=10)
    p.add_argument('--synthetic-structured', type=int, default=1,
                   help='--synthetic: 1 (default) = class-structured images (imagenet_loading.SyntheticImageNet(structured=True)) and '
                        'the classifier\'s last layer fitted to the training split in closed form (zoo.fit_centroid_head), so that the '
                        'pipeline reports real fooling rates without pretrained weights; 0 = U[0,1) noise images, random-init head '
                        '(performance.py\'s correctly-classified filter then keeps next to nothing)')
    p.add_argument('--image-size', type=int, default=224)
    p.add_argument('--n-atoms', type=int, default=100)
    p.add_argument('--steps', type=int, default=500)
    p.add_argument('--batch-size', type=int, default=100)
    p.add_argument('--loss', default='logits', choices=['logits', 'ce'])
    p.add_argument('--method', default='gd', choices=['gd', 'alter'])
    p.add_argument('--dtype', default='fp32', choices=['fp32', 'bf16'])
    p.add_argument('--cache-labels', type=int, default=1,
                   help='1 (default): compute the clean pseudo-label of a training image once instead of in every epoch '
                        '(one classifier forward less per learning step; measured result-neutral); 0 = the reference op sequence')
    p.add_argument('--val-every', type=int, default=1,
                   help='validate every this many epochs (0 = after the last epoch only); the dictionary file is the same')
    p.add_argument('--upload-workers', type=int, default=0,
                   help='worker processes fetching the dataset items for the one-time upload into HBM (JPEG decoding)')
    p.add_argument('--clean-accuracy', type=int, default=1,
                   help='1 (default, as upstream): print the clean top-1 accuracy of the classifier over the whole dataset '
                        'first (model_accuracy.py); 0 skips that pass')
    p.add_argument('--fast-classifier', type=int, default=1,
                   help='bf16 ResNets: run the frozen classifier on the hand-written stem / pointwise / 3x3 kernels '
                        '(zoo.FusedResNet, same function up to bf16 rounding); 0 = plain PyTorch modules')
    return p


def spawn_ranks(args):
    """--distributed outside a torchrun environment: start one rank per visible GPU ourselves.  The parent makes no GPU
    call (device_count() does not initialise HIP on this stack); the ranks are fresh children of torch.distributed.run.
    Replaces the SLURM bootstrap of env_setting.py:10-28."""
    import socket
    import subprocess
    import sys
    n = torch.cuda.device_count()
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={max(1, n)}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # in place before a rank's first HIP call, whatever the rank's script does first
    return subprocess.run(cmd, env=env).returncode


def main(args):
    under_srun = int(os.environ.get("SLURM_NTASKS", "1")) > 1 and "SLURM_PROCID" in os.environ      # the reference's launch form
    if args.distributed and "WORLD_SIZE" not in os.environ and not under_srun and torch.cuda.device_count() > 1:
        raise SystemExit(spawn_ranks(args))
    if args.distributed:
        from dl_attack_on_imagenet_amd import dist as adist
        os.environ.setdefault(adist.IPC_ENV, "0")       # before this process's first HIP call (see dist.init_from_env)
    if not torch.cuda.is_available():
        print('Check cuda setting for model training on ImageNet')       # demo_dL_attack.py:30-32
        return
    if not args.distributed:
        torch.cuda.set_device(args.gpu)
        device = torch.device('cuda', args.gpu)
    else:
        # one process per GPU (torchrun env); the process group also shards the evaluation below over the ranks
        from dl_attack_on_imagenet_amd import dist as adist
        _, _, local_rank = adist.init_from_env(slurm=True)        # torchrun's variables, else srun's (env_setting.py:7-16)
        device = torch.device('cuda', adist.local_device_index(local_rank))
        torch.cuda.set_device(device)

    model_name = args.model.lower()          # names the dictionary file, as upstream (demo_dL_attack.py:41, adil.py:89-91)
    dtype = torch.bfloat16 if args.dtype == 'bf16' else torch.float32
    fast = bool(args.fast_classifier) and dtype == torch.bfloat16 and zoo.canonical_name(model_name).startswith('resnet')

    structured = bool(args.synthetic and args.synthetic_structured)
    if args.synthetic:
        dataset = SyntheticImageNet(num_classes=args.synthetic_classes, size=args.image_size, seed=args.seed,
                                    structured=structured)
        n_classes = min(args.trained_classes, args.synthetic_classes)
    else:
        dataset, _ = load_ImageNet()
        n_classes = args.trained_classes
    train_dataset, val_dataset, test_dataset = dataset_split_by_class(
        dataset, [args.num_train_per_class, 2, 5], number_of_classes=n_classes)         # demo_dL_attack.py:69-78

    weights = args.weights
    if structured and weights is None:
        weights = _fitted_weights(model_name, args.seed, train_dataset, n_classes, device)
    model = zoo.build_classifier(model_name, seed=args.seed, weights=weights, device=device, dtype=dtype,
                                 channels_last=fast, fuse_bn_act=fast, fuse_stem=fast,
                                 head_fp32="inference" if fast else False)      # fp32 logits inside the DDrague inference loop
    if args.clean_accuracy:                                                               # demo_dL_attack.py:65-66
        from model_accuracy import model_accuracy, model_accuracy_distributed
        dataset.indexed = False
        acc = (model_accuracy_distributed(dataset, model, device) if torch.distributed.is_initialized()
               else model_accuracy(dataset, model, device=device))
        print("accuracy of the the model {} is {}".format(model_name, float(acc) * 100))
    val_loader = torch.utils.data.DataLoader(val_dataset, batch_size=10, shuffle=False)
    test_loader = torch.utils.data.DataLoader(test_dataset, batch_size=20, shuffle=False)

    eps, norm = 8 / 255, 'linf'
    attacks_hyper = {
        'adil': perf.get_atks(model, ADIL, 'n_atoms', [args.n_atoms], 'kappa', [50], alpha=0 / 255,
                              data_train=train_dataset, norm=norm, attack='supervised', eps=eps, steps=args.steps,
                              targeted=False, step_size=0.01, batch_size=args.batch_size, model_name=model_name,
                              is_distributed=args.distributed, steps_in=1, loss=args.loss, method=args.method,
                              data_val=val_dataset, warm_start=False, steps_inference=args.steps_inference,
                              stream_dtype=dtype if dtype != torch.float32 else None,
                              cache_labels=bool(args.cache_labels), val_every=args.val_every,
                              upload_workers=args.upload_workers),
    }
    out_dir = 'dict_model_ImageNet_version_constrained'
    os.makedirs(out_dir, exist_ok=True)                                                   # quirk Q14: upstream assumes it exists
    print('Evaluation process')
    writer = int(os.environ.get('RANK', '0')) == 0                                        # one writer under --distributed
    val_perf = perf.get_performance(attacks_hyper, model, _cast_loader(val_loader, dtype), device=device)
    if writer:                                                                            # demo_dL_attack.py:148-151
        torch.save(val_perf, os.path.join(out_dir, f'model_sampling_adil_inference_rlts_sampling_'
                                                   f'{args.num_train_per_class * args.trained_classes}_'
                                                   f'{args.steps_inference}_{args.seed}_ce.bin'))
    print('Test process')
    test_perf = perf.get_performance(attacks_hyper, model, _cast_loader(test_loader, dtype), device=device)
    if writer:                                                                            # demo_dL_attack.py:153-156
        torch.save(test_perf, os.path.join(out_dir, 'model_adil_resultat_test_ce.bin'))
    return val_perf, test_perf


def _fitted_weights(model_name, seed, train_dataset, n_classes, device):
    """No pretrained weights offline: fit the last layer of the seeded random-weight network to the structured training
    split (zoo.fit_centroid_head, closed form) and hand the result to build_classifier as a local state_dict — the same
    route a torchvision checkpoint takes.  Every rank of a --distributed run computes the same file content."""
    import tempfile
    plain = zoo.build_classifier(model_name, seed=seed, device=device)
    train_dataset.indexed = False
    images = torch.stack([train_dataset[i][0] for i in range(len(train_dataset))])
    labels = torch.tensor([int(train_dataset[i][1]) for i in range(len(train_dataset))])
    margins, pred = zoo.fit_centroid_head(plain, images, labels, n_classes, device)
    print(f'fitted the classifier head to {len(labels)} structured training images: accuracy '
          f'{float((pred == labels).float().mean()):.3f}, median clean margin {float(margins.median()):.1f}')
    path = os.path.join(tempfile.mkdtemp(prefix='adil_demo_'), f'{model_name}_fitted.pt')
    torch.save(plain[-1].state_dict(), path)
    return path


class _CastLoader:
    """A DataLoader whose image batches come out in `dtype`; everything else (`batch_size`, `dataset`, `len`) is the
    wrapped loader's, which performance.py relies on (the classifier's batch bucket, the transfer evaluation's
    `len(data.dataset)`, performance.py:130 / :207 here)."""

    def __init__(self, loader, dtype):
        self._loader, self._dtype = loader, dtype

    def __iter__(self):
        for x, y in self._loader:
            yield x.to(self._dtype), y

    def __len__(self):
        return len(self._loader)

    def __getattr__(self, name):
        return getattr(self._loader, name)


def _cast_loader(loader, dtype):
    return loader if dtype == torch.float32 else _CastLoader(loader, dtype)


if __name__ == '__main__':
    args = build_parser().parse_args()
    print(args.seed)
    torch.random.manual_seed(args.seed)                                                   # demo_dL_attack.py:209-211
    random.seed(args.seed)
    np.random.seed(args.seed)
    main(args)
