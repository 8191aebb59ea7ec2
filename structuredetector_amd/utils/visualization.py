"""Drawing of decoded annotations (debug / `detect` output; reference: src/sdnet/utils/visualization.py:6-50,173-193).
Host-side PIL code: nothing here is on the hot path."""
from __future__ import annotations

import numpy as np
import torch
from PIL import Image, ImageDraw

_MEAN = (0.485, 0.456, 0.406)
_STD = (0.229, 0.224, 0.225)


def un_normalize(tensor: torch.Tensor) -> torch.Tensor:
    """visualization.py:6-10: undo the ImageNet normalisation of a (B,3,H,W) or (3,H,W) tensor."""
    mean = torch.tensor(_MEAN, device=tensor.device)[..., None, None]
    std = torch.tensor(_STD, device=tensor.device)[..., None, None]
    return tensor * std + mean


def _to_pil(image, unnorm_image=True) -> Image.Image:
    if isinstance(image, torch.Tensor):
        img = un_normalize(image) if unnorm_image else image
        arr = (img.detach().float().cpu().clamp(0, 1) * 255).round().to(torch.uint8).permute(1, 2, 0).numpy()
        return Image.fromarray(np.ascontiguousarray(arr))
    return image.copy()


def _dot(draw, x, y, r, color):
    draw.ellipse([x - r, y - r, x + r, y + r], fill=color, outline=color)


def draw(image, annotation, args, unnorm_image=True) -> Image.Image:
    """visualization.py:13-50: anchors as discs in the label colour, parts in the part colour, a white link per part.
    Radius and line width are 1 % of the shorter image side."""
    img = _to_pil(image, unnorm_image)
    pen = ImageDraw.Draw(img)
    size = int(min(img.size) * 1 / 100)
    for obj in annotation.objects:
        for kp in obj.parts:
            pen.line([obj.x, obj.y, kp.x, kp.y], fill="white", width=size)
            _dot(pen, kp.x, kp.y, size, args._part_color_map[kp.kind])
        _dot(pen, obj.x, obj.y, size, args._label_color_map[obj.name])
    return img


def draw_keypoints(image, keypoints, args) -> Image.Image:
    """visualization.py:173-193: loose keypoints (label or part kinds)."""
    img = image.copy()
    pen = ImageDraw.Draw(img)
    size = int(min(img.size) * 1 / 100)
    for kp in keypoints:
        if kp.kind in args.labels:
            color = args._label_color_map[kp.kind]
        elif kp.kind in args.parts:
            color = args._part_color_map[kp.kind]
        else:
            raise ValueError(f"unknown keypoint kind {kp.kind!r}")
        _dot(pen, kp.x, kp.y, size, color)
    return img
