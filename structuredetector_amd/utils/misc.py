"""Small host helpers kept for API parity with src/sdnet/utils/utils.py:324-338,364-381."""
from pathlib import Path

import numpy as np
import torch


def mkdir_if_needed(directory):
    Path(directory).mkdir(exist_ok=True)


def set_seed(seed=1975846251):
    torch.manual_seed(seed)
    np.random.seed(seed % (2**32))
    if torch.cuda.is_available():
        torch.cuda.manual_seed_all(seed)


def clip_annotation(annotation, img_size):
    """utils.py:364-381: clamp every coordinate to [0, size-1], in place."""
    w, h = img_size
    cx = lambda v: min(max(v, 0), w - 1)
    cy = lambda v: min(max(v, 0), h - 1)
    for obj in annotation.objects:
        obj.x, obj.y = cx(obj.x), cy(obj.y)
        for p in obj.parts:
            p.x, p.y = cx(p.x), cy(p.y)
        if obj.box is not None:
            b = obj.box
            b.x_min, b.x_max, b.y_min, b.y_max = cx(b.x_min), cx(b.x_max), cy(b.y_min), cy(b.y_max)
    return annotation


def hflip_annotation(annotation, img_size):
    """utils.py:384-398: mirror every x about the image width, in place (boxes keep x_min <= x_max)."""
    w, _ = img_size
    for obj in annotation.objects:
        obj.x = w - obj.x - 1
        for p in obj.parts:
            p.x = w - p.x - 1
        if obj.box is not None:
            obj.box.x_min, obj.box.x_max = w - obj.box.x_max - 1, w - obj.box.x_min - 1
    return annotation


def vflip_annotation(annotation, img_size):
    """utils.py:401-415."""
    _, h = img_size
    for obj in annotation.objects:
        obj.y = h - obj.y - 1
        for p in obj.parts:
            p.y = h - p.y - 1
        if obj.box is not None:
            obj.box.y_min, obj.box.y_max = h - obj.box.y_max - 1, h - obj.box.y_min - 1
    return annotation


def get_unique_color_map(labels):
    """utils.py:476-479: a stable RGB triple per name (first three bytes of its xxh64 digest)."""
    from xxhash import xxh64_digest
    return {n: (*xxh64_digest(n.encode())[:3],) for n in labels}


def files_with_extension(folder, extension: str):
    """utils.py:327-328."""
    return [f for f in Path(folder).iterdir() if f.suffix == extension]


class AverageMeter:
    """Running mean of the values passed to `update` (reference: src/sdnet/utils/utils.py:311-324; same attributes `sum`, `count`, `avg`)."""

    def __init__(self):
        self.reset()

    def reset(self):
        self.sum, self.count, self.avg = 0.0, 0, 0.0

    def update(self, value):
        self.sum += value
        self.count += 1
        self.avg = self.sum / self.count
        return self.avg


def dict_grouping(iterable, key):
    """{key(element): [elements in input order]} (reference: src/sdnet/utils/utils.py:470-474; used by its Evaluator to split by label / kind)."""
    groups = {}
    for element in iterable:
        groups.setdefault(key(element), []).append(element)
    return groups

