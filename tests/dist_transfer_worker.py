"""Worker of tests/test_gpu_adil.py::test_transfer_evaluation_data_parallel: run under torch.distributed.run, every rank
evaluates its share of the G15 loader's batches with the product's performance.get_transfer_performance; rank 0 writes
the result as JSON to argv[1]."""
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, HERE)
import numpy as np
import torch

import performance as perf
from attacks import ADIL
from dl_attack_on_imagenet_amd import dist as adist
from tinynet import tinynet_from_npz

rank, world, local_rank = adist.init_from_env()
dev = torch.device("cuda", adist.local_device_index(local_rank))
torch.cuda.set_device(dev)
z = np.load(os.path.join(HERE, "golden", "g15_transfer.npz"))
t = lambda a: torch.from_numpy(np.asarray(a))
targets = {name: tinynet_from_npz(z, prefix=f"{name}.").to(dev) for name in ("src", "t1", "t2")}
dict_dir = sys.argv[2]
if rank == 0:
    torch.save([t(z["d"]), torch.zeros(1), [], [], torch.tensor(0.)], os.path.join(dict_dir, "ImageNet_g15.bin"))
torch.distributed.barrier()
atk = ADIL(targets["src"], eps=float(z["eps"]), n_atoms=z["d"].shape[-1], attack="supervised", model_name="g15", loss="logits",
           steps_inference=int(z["steps"]), kappa=float(z["kappa"]), dict_dir=dict_dir)
loader = torch.utils.data.DataLoader(torch.utils.data.TensorDataset(t(z["images"]), t(z["labels"])),
                                     batch_size=int(z["batch_size"]), shuffle=False)
out = perf.get_transfer_performance({"adil": [atk]}, targets, loader, device=dev)
if rank == 0:
    json.dump({"world": world, "perf": out["adil"]}, open(sys.argv[1], "w"))
torch.distributed.barrier()
torch.distributed.destroy_process_group()
