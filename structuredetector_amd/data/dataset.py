"""Batch assembly helpers.  `collate_fn` keeps the reference's contract
(src/sdnet/data/dataset.py:58-87): stack every Encode field along a new leading dim, keep
`annotation` as a list."""
import torch

_TENSOR_KEYS = ["image", "anchor_hm", "part_hm", "anchor_offsets", "part_offsets", "embeddings", "anchor_inds",
                "part_inds", "anchor_mask", "part_mask"]


def collate_fn(elements):
    batch = {k: torch.stack([e[k] for e in elements], dim=0) for k in _TENSOR_KEYS}
    batch["annotation"] = [e["annotation"] for e in elements]
    return batch


# ---------------------------------------------------------------------------------------------
# Minimal directory reader (callers of the hot path; SURVEY.md 8f-2 marks the full augmentation pipeline "next").
# Same on-disk format as the reference's CropDataset (src/sdnet/data/dataset.py:13-49): one JSON per image
# with {"image_path", "img_size", "objects": [...]}; images are resized to (width, height) and normalised with
# the ImageNet statistics (ValidationAugmentation, src/sdnet/data/transforms.py:255-261).  No flips / jitter.
# ---------------------------------------------------------------------------------------------
_MEAN = (0.485, 0.456, 0.406)
_STD = (0.229, 0.224, 0.225)


class CropDataset:
    """raw=False: items are (normalised (3, height, width) fp32 CPU tensor, annotation in network-input pixels) -- resize and
    normalisation on the host with PIL (the reference's ValidationAugmentation minus Encode).
    raw=True: items are ((H, W, 3) uint8 CPU tensor, annotation in ORIGINAL pixels): decode only; resize / flips / normalisation
    then run for the whole batch on the GPU (data/augment.py)."""

    def __init__(self, args, directories, raw=False):
        from pathlib import Path
        dirs = [directories] if isinstance(directories, (str, Path)) else list(directories)
        self.args = args
        self.raw = raw
        self.files = sorted(f for d in dirs for f in Path(d).iterdir() if f.suffix == ".json")

    def __len__(self):
        return len(self.files)

    def __getitem__(self, index):
        import numpy as np
        from PIL import Image

        from ..utils.misc import clip_annotation
        from ..utils.types import ImageAnnotation
        ann = ImageAnnotation.from_json(self.files[index], self.args.anchor_name)
        path = ann.image_path if ann.image_path.is_absolute() else self.files[index].parent / ann.image_path.name
        img = Image.open(path).convert("RGB")
        ann.img_size = img.size
        if self.raw:
            return torch.from_numpy(np.asarray(img, np.uint8).copy()), ann
        W, H = self.args.width, self.args.height
        ann.resize(img.size, (W, H))                                   # Resize transform, transforms.py:47-60
        # the reference's pipeline ends with Encode, which clips the annotation IN PLACE to the network input
        # (transforms.py:154, utils.py:364-381) before the Evaluator / the loss see it
        clip_annotation(ann, (W, H))
        arr = np.asarray(img.resize((W, H), Image.BILINEAR), np.float32) / 255.0
        arr = (arr - np.asarray(_MEAN, np.float32)) / np.asarray(_STD, np.float32)
        return torch.from_numpy(arr).permute(2, 0, 1).contiguous(), ann


class PredictionDataset:
    """src/sdnet/data/dataset.py:166-181 + PredictionTransformation (transforms.py:265-280): every `.jpg` of a directory.
    raw=False: items are {"img": (3,H,W) tensor resized to the network input and ImageNet-normalised on the host, "img_size": (w, h) of
    the original}.  raw=True: items are ((H, W, 3) uint8 CPU tensor, empty ImageAnnotation carrying image_path and img_size): decode
    only; Resize + Normalize then run for the whole batch on the GPU (model/predictor.py)."""

    def __init__(self, directory, args, raw=False):
        from pathlib import Path
        self.images = sorted(f for f in Path(directory).iterdir() if f.suffix == ".jpg")
        self.args = args
        self.raw = raw

    def __len__(self):
        return len(self.images)

    def __getitem__(self, index):
        import numpy as np
        from PIL import Image
        if index >= len(self.images):
            raise IndexError(index)
        img = Image.open(self.images[index]).convert("RGB")
        size = img.size
        if self.raw:
            from ..utils.types import ImageAnnotation
            return torch.from_numpy(np.asarray(img, np.uint8).copy()), ImageAnnotation(self.images[index], [], img_size=size)
        arr = np.asarray(img.resize((self.args.width, self.args.height), Image.BILINEAR), np.float32) / 255.0
        arr = (arr - np.asarray(_MEAN, np.float32)) / np.asarray(_STD, np.float32)
        return {"img": torch.from_numpy(arr).permute(2, 0, 1).contiguous(), "img_size": size}
