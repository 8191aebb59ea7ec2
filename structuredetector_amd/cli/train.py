"""`train` entry point (reference: src/sdnet/cli/train.py:5-13).  Multi-GPU: launch one process per GPU with
`python -m torch.distributed.run --nproc-per-node N -m structuredetector_amd.cli.train ...`."""
import os

import torch
import torch.distributed as dist

from ..model.trainer import Trainer
from ..utils import Arguments


def main(argv=None):
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world > 1:
        torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", "0")))
        dist.init_process_group("nccl")
    args = Arguments().parse(argv)
    assert args.synthetic or args.train_dir, "Path to a directory with train samples must be specified."
    Trainer(args).train()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
