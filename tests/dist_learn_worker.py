"""Worker of tests/test_gpu_adil.py::test_two_rank_* : run under torch.distributed.run with 2 ranks, each rank runs the
product's data-parallel learner (ADIL.learn_dictionary_distributed) on a small seeded problem — ragged shards, explicit
global batches incl. one that a single rank owns alone, sharded validation with the global stop test — and writes what
the parent test compares: its final dictionary (must be bit-identical across ranks), the backend it ran on, the
reducer's timing record; rank 0 writes the reference's dictionary file.

Two launch modes, same code: RCCL with one GPU per rank (needs >= 2 visible GPUs), or the one-GPU rehearsal
(ADIL_DIST_BACKEND=gloo ADIL_SHARE_GPU=1)."""
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, HERE)
from dl_attack_on_imagenet_amd import dist as adist           # noqa: E402  (before anything touches the GPU)

rank, world, local_rank = adist.init_from_env()
import torch                                                   # noqa: E402

from attacks import ADIL                                       # noqa: E402
from dist_learn_problem import problem, IndexedImages          # noqa: E402

out_dir = sys.argv[1]
dev = torch.device("cuda", adist.local_device_index(local_rank))
torch.cuda.set_device(dev)
p = problem()
net = p["net"].to(dev)
atk = ADIL(net, data_train=None, model_name="dp2", dict_dir=os.path.join(out_dir, "dicts"), **p["kw"])
learner = atk.learn_dictionary_distributed(IndexedImages(p["images"]), IndexedImages(p["val"]))
torch.save(learner.d.cpu(), os.path.join(out_dir, f"d_rank{rank}.pt"))
torch.cuda.synchronize(dev)                                    # the reducer's event brackets are read below
info = learner.reducer.describe()
info.update(rank=rank, device=str(dev), device_name=torch.cuda.get_device_name(dev), visible_gpus=torch.cuda.device_count(),
            async_work=bool(getattr(learner.reducer, "saw_async_work", False)))
json.dump(info, open(os.path.join(out_dir, f"info_rank{rank}.json"), "w"))
torch.distributed.barrier()
torch.distributed.destroy_process_group()
