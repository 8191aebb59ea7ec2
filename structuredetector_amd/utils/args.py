"""Command-line surface of the `train` / `evaluate` entry points.

Same flags, short options, dest names, defaults, validation and derived fields as the
reference's `Arguments` (src/sdnet/utils/args.py:17-269; table in SURVEY.md A.4), declared as a
table.  Extra (build-only) flags are grouped at the end and all default to "off".
"""
from __future__ import annotations

import argparse
import json
from multiprocessing import cpu_count
from pathlib import Path

import torch

from .misc import get_unique_color_map, set_seed

# (flags, kwargs)
_FLAGS = [
    (("--train_dir",), dict(type=str, help="The training directory.")),
    (("--valid_dir",), dict(type=str, help="The validation directory.")),
    (("--labels", "-m"), dict(type=str, default="labels.json", help="Json file of anchor and part names.")),
    (("--anchor_name", "-s"), dict(type=str, default="anchor", help="Name of the keypoint representing the anchor.")),
    (("--width", "-W"), dict(type=int, default=512, help="The network input width.")),
    (("--height", "-H"), dict(type=int, default=512, help="The network input height.")),
    (("--in_channels", "-c"), dict(type=int, default=3, help="Number of input channels.")),
    (("--fpn_depth",), dict(type=int, default=128, help="Depth of FPN layers of the decoder.")),
    (("--load_model", "-o"), dict(default=None, dest="pretrained_model", help="Load a previously trained model.")),
    (("--batch_size", "-b"), dict(type=int, default=8, help="Batch size for training.")),
    (("--epochs", "-e"), dict(type=int, default=100, help="The number of epochs to train.")),
    (("--no_augmentation", "-a"), dict(action="store_true", help="Disable training augmentations.")),
    (("--learning_rate", "-l"), dict(type=float, default=1e-3, help="The learning rate for training.")),
    (("--lr_step",), dict(type=int, default=3, help="Number of divisions by 10 of the learning rate (0 = off).")),
    (("--down_ratio", "-g"), dict(type=float, default=4.0, help="Downsampling ratio of the network.")),
    (("--hm_loss_fn", "-f"), dict(type=str, default="mse", help="Heatmap loss: 'focal' or 'mse'.")),
    (("--max_objects", "-n"), dict(type=int, default=20, help="Maximum number of objects per image.")),
    (("--max_parts", "-k"), dict(type=int, default=40, help="Maximum number of parts per image.")),
    (("--hm_weight",), dict(type=float, default=1.0, help="Weight for the heatmap loss.")),
    (("--offset_weight",), dict(type=float, default=0.001, help="Weight for the offset loss.")),
    (("--embedding_weight",), dict(type=float, default=0.001, help="Weight for the embedding loss.")),
    (("--sigma_gauss",), dict(type=float, default=10 / 100, help="Gaussian size, fraction of the image side.")),
    (("--conf_threshold", "-t"), dict(type=float, default=50 / 100, help="Confidence threshold in [0, 1].")),
    (("--dist_threshold", "-d"), dict(type=float, default=5 / 100, help="Evaluation radius, fraction of min side.")),
    (("--decoder_dist_thresh",), dict(type=float, default=10 / 100, help="Linkage radius, fraction of min side.")),
    (("--csi_threshold",), dict(type=float, default=75 / 100, help="Threshold on the CSI metric.")),
    (("--save_csv_eval",), dict(dest="csv_path", type=Path)),
    (("--amp",), dict(action="store_true", dest="use_amp", help="Automatic mixed precision.")),
    # build-only additions (data-parallel launcher / synthetic input)
    (("--synthetic",), dict(type=int, default=0, help="Train/evaluate on N seeded synthetic scenes instead of a directory.")),
    (("--steps",), dict(type=int, default=0, help="Stop training after this many optimizer steps (0 = full epochs).")),
    (("--bf16_inference",), dict(action="store_true", help="Eval-mode forward on the bf16 backbone (fp32 decode); training unaffected.")),
    (("--resume",), dict(type=str, default=None, help="Continue a run from trainings/<stamp>/resume.pth (weights, Adam state, scheduler, epoch).")),
    (("--decode_workers",), dict(type=int, default=0, help="Image decode threads of the directory feed (0 = half of this rank's CPU share, 2 .. 16).")),
    (("--prefetch",), dict(type=int, default=3, help="Batches decoded and uploaded ahead of the training step.")),
    (("--backbone_weights",), dict(type=str, default=None, help="torchvision ResNet-34 ImageNet state_dict (resnet34-b627a593.pth) for "
                                   "the trunk: what the reference downloads for pretrained=True (default: $SDNET_BACKBONE_WEIGHTS, then the torch hub cache).")),
    (("--log_dir",), dict(type=str, default=None, help="Directory for the training scalars (default: the run's save directory).")),
    (("--eval_batch",), dict(type=int, default=16, help="Images per forward + decode launch in evaluate / validation / detect.")),
]

_POSITIVE = ["in_channels", "fpn_depth", "batch_size", "epochs", "learning_rate", "down_ratio", "max_objects", "max_parts"]
_NON_NEGATIVE = ["lr_step", "hm_weight", "offset_weight", "embedding_weight"]
_UNIT = ["conf_threshold", "dist_threshold", "decoder_dist_thresh", "csi_threshold"]


def _name_map(value):
    if isinstance(value, dict):
        return value
    if isinstance(value, list):
        return {name: i for i, name in enumerate(value)}
    return {value: 0}


def finalize(args):
    """Validation + derived fields (args.py:178-269) on an already parsed namespace."""
    for side in ("width", "height"):
        v = getattr(args, side)
        assert v % 32 == 0 and v > 0, f"{side.capitalize()} should be divisible by 32 and greater than 0"
    for k in _POSITIVE:
        assert getattr(args, k) > 0, f"'{k}' should be greater than 0"
    for k in _NON_NEGATIVE:
        assert getattr(args, k) >= 0, f"'{k}' should be greater than or equal to 0"
    for k in _UNIT:
        assert 0 <= getattr(args, k) <= 1, f"'{k}' should be in [0.0, 1.0]"
    assert 0 < args.sigma_gauss <= 1, "'sigma_gauss' should be in ]0.0, 1.0]"

    args.lr_step = int(args.epochs / args.lr_step) if args.lr_step != 0 else args.epochs
    for k in ("train_dir", "valid_dir", "pretrained_model"):
        if getattr(args, k, None) is not None:
            setattr(args, k, Path(getattr(args, k)).expanduser().resolve())

    if not isinstance(args.labels, dict):
        names = json.loads(Path(args.labels).expanduser().resolve().read_text())
        args.labels = _name_map(names["labels"])
        args.parts = _name_map(names["parts"])

    args.use_cuda = torch.cuda.is_available()          # ROCm PyTorch reports "cuda"
    if not args.use_cuda:
        raise RuntimeError("structuredetector_amd needs an MI355X visible as torch device 'cuda' (no CPU path)")
    args.device = torch.device("cuda", torch.cuda.current_device())
    args.num_workers = min(cpu_count(), 4)          # (args.py:251: the reference's DataLoader workers; the decode threads here: --decode_workers)
    set_seed(926354916)

    if args.hm_loss_fn.lower() not in {"focal", "mse"}:
        raise IOError(f"'hm_loss_fn' should either be 'focal' or 'mse', not {args.hm_loss_fn}.")
    args._r_labels = {v: k for k, v in args.labels.items()}
    args._r_parts = {v: k for k, v in args.parts.items()}
    args._label_color_map = get_unique_color_map(args.labels)          # args.py:264-267
    args._part_color_map = get_unique_color_map(args.parts)
    return args


class Arguments:
    def __init__(self):
        self.parser = argparse.ArgumentParser(formatter_class=argparse.ArgumentDefaultsHelpFormatter)
        for flags, kw in _FLAGS:
            self.parser.add_argument(*flags, **kw)

    def parse(self, argv=None):
        return finalize(self.parser.parse_args(argv))
