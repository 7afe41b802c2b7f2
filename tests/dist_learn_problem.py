"""The seeded problem shared by tests/dist_learn_worker.py (2 ranks) and its parent test (single process)."""
import torch

from tinynet import make_tinynet

N_IMG, N_VAL, K, EPS = 23, 9, 5, 0.3          # 23 = 12 + 11, 9 = 5 + 4: ragged shards
# explicit GLOBAL batches per epoch; epoch 0's second batch is owned by rank 0 alone (rows < 12), its third by rank 1 alone
EPOCH_BATCHES = [[[0, 13, 5, 20, 7, 15, 2, 22], [1, 3, 4, 6, 8, 9, 10, 11], [12, 14, 16, 17, 18, 19, 21]],
                 [[22, 1, 12, 3, 14, 5, 16, 7], [18, 9, 20, 11, 0, 13, 2, 15], [4, 17, 6, 19, 8, 21, 10]]]
VAL_BATCHES = [[[0, 5, 1, 6, 2], [7, 3, 8, 4]], [[0, 1, 2, 3, 4], [5, 6, 7, 8]]]       # [0,1,2,3,4]: rank 0 alone


class IndexedImages(torch.utils.data.Dataset):
    """The reference's `indexed` protocol (imagenet_loading.py:8-18)."""

    def __init__(self, images):
        self.images, self.indexed = images, False

    def __len__(self):
        return len(self.images)

    def __getitem__(self, i):
        return (i, self.images[i], 0) if self.indexed else (self.images[i], 0)


def problem():
    g = torch.Generator().manual_seed(1234)
    images = torch.rand(N_IMG, 3, 32, 32, generator=g)
    val = torch.rand(N_VAL, 3, 32, 32, generator=g)
    d0 = -1 + 2 * torch.rand(3, 32, 32, K, generator=g)
    v0 = torch.rand(N_IMG, K, generator=g)
    kw = dict(eps=EPS, steps=len(EPOCH_BATCHES), n_atoms=K, batch_size=8, loss="ce", step_size=0.01, kappa=50.0,
              init_d=d0, init_v=v0, epoch_batches=EPOCH_BATCHES, val_batches=VAL_BATCHES)
    return dict(net=make_tinynet(31), images=images, val=val, kw=kw)
