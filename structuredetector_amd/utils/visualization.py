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


def _palette(color_map, names_by_index, n, device):
    """(n, 3) colours of channels 0 .. n-1; channels without a name are black (visualization.py:60-73)."""
    rows = [color_map.get(names_by_index.get(i), (0, 0, 0)) for i in range(n)]
    return torch.tensor(rows, device=device).reshape(n, 3)


def draw_heatmaps(anchor_hm: torch.Tensor, part_hm: torch.Tensor, args):
    """visualization.py:53-91: per pixel the colour of the strongest channel scaled by its value; two (3, h, w) uint8 tensors (truncation,
    first maximal channel on ties).  One sample, not a batch."""
    if anchor_hm.dim() != 3 or part_hm.dim() != 3:
        raise AssertionError("Do not send batched data to this function, only one sample")

    def shade(hm, palette):
        top, which = hm.max(dim=0)                                  # (h, w): value and channel of the strongest map
        rgb = palette[which].permute(2, 0, 1).float()               # (3, h, w)
        return (rgb * top).to(torch.uint8)

    return (shade(anchor_hm, _palette(args._label_color_map, args._r_labels, anchor_hm.shape[0], anchor_hm.device)),
            shade(part_hm, _palette(args._part_color_map, args._r_parts, part_hm.shape[0], part_hm.device)))


def draw_kp_and_emb(image, topk_obj, topk_kp, embeddings, args) -> Image.Image:
    """visualization.py:94-148: the decoder's raw peaks on the un-normalised image -- anchors and parts scoring at least `args.conf_threshold`
    as discs in their colour, every part with a line along its embedding vector (output pixels x down_ratio).  Batch of one."""
    img = _to_pil(image, True)
    pen = ImageDraw.Draw(img)
    size = int(min(img.size) * 1 / 100)
    ratio, thresh = args.down_ratio, args.conf_threshold
    scores, _, classes, ys, xs = (t.reshape(-1).tolist() for t in topk_obj)
    for score, cls, y, x in zip(scores, classes, ys, xs):
        if score >= thresh:
            _dot(pen, x * ratio, y * ratio, size, args._label_color_map[args._r_labels[int(cls)]])
    scores, _, classes, ys, xs = (t.reshape(-1).tolist() for t in topk_kp)
    vecs = embeddings.reshape(-1, 2).tolist()
    for score, cls, y, x, (ex, ey) in zip(scores, classes, ys, xs, vecs):
        if score < thresh:
            continue
        color = args._part_color_map[args._r_parts[int(cls)]]
        px, py = x * ratio, y * ratio
        _dot(pen, px, py, size, color)
        pen.line([px, py, px + ratio * ex, py + ratio * ey], fill=color, width=size)
    return img


def draw_embeddings(image, embeddings: torch.Tensor, args) -> Image.Image:
    """visualization.py:151-170: the embedding field of one image as red segments, one per fourth output pixel in each direction, from the
    pixel's position in the input image along its (x, y) vector."""
    if embeddings.shape[0] != 1:
        raise AssertionError("BS should be one")
    img = _to_pil(image, True)
    pen = ImageDraw.Draw(img)
    width = int(min(img.size) * 0.5 / 100)
    ratio = args.down_ratio
    field = (embeddings[0] * ratio).detach().float().cpu()[:, ::4, ::4]          # (2, ceil(h / 4), ceil(w / 4)) in input pixels
    for j, row in enumerate(field.permute(1, 2, 0).tolist()):
        for i, (dx, dy) in enumerate(row):
            x1, y1 = 4 * i * ratio, 4 * j * ratio
            pen.line([x1, y1, x1 + dx, y1 + dy], fill=(255, 0, 0), width=width)
    return img

