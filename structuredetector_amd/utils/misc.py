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
