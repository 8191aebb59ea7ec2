"""Seeded synthetic scenes: n objects per image, anchors uniform over the image, 1-3 parts per
object within +-40 px, delivered directly as the flat arrays `Encode.plan` consumes (no Python objects)."""
from __future__ import annotations

import numpy as np


def synthetic_batch(rng: np.random.Generator, B, img_w, img_h, M, N, n_min=6, n_max=12, parts_min=1, parts_max=3, spread=40.0):
    n_obj = rng.integers(n_min, n_max + 1, size=B)
    n = int(n_obj.sum())
    o_xy = np.stack([rng.uniform(0, img_w - 1, n), rng.uniform(0, img_h - 1, n)], axis=1)
    o_lab = rng.integers(0, M, size=n)
    o_np = rng.integers(parts_min, parts_max + 1, size=n)
    m = int(o_np.sum())
    owner = np.repeat(np.arange(n), o_np)
    p_xy = o_xy[owner] + rng.uniform(-spread, spread, size=(m, 2))
    p_xy[:, 0] = np.clip(p_xy[:, 0], 0, img_w - 1)
    p_xy[:, 1] = np.clip(p_xy[:, 1], 0, img_h - 1)
    p_kind = rng.integers(0, N, size=m)
    return (n_obj.astype(np.int64), o_lab.astype(np.int64), o_xy, o_np.astype(np.int64), p_kind.astype(np.int64), p_xy)


def synthetic_samples(args, n, seed=926354916):
    """n single-image samples (normalised-noise image on args.device, ImageAnnotation with img_size) for evaluate / validation."""
    import torch

    from ..utils.types import ImageAnnotation, Keypoint, Object
    rng = np.random.default_rng(seed)
    gen = torch.Generator(device=args.device).manual_seed(seed)
    for i in range(n):
        n_obj, o_lab, o_xy, o_np, p_kind, p_xy = synthetic_batch(rng, 1, args.width, args.height, len(args.labels), len(args.parts))
        objs, j = [], 0
        for k in range(int(n_obj[0])):
            parts = [Keypoint(args._r_parts[int(p_kind[j + q])], *p_xy[j + q]) for q in range(int(o_np[k]))]
            j += int(o_np[k])
            objs.append(Object(args._r_labels[int(o_lab[k])], Keypoint(args.anchor_name, *o_xy[k]), parts))
        yield (torch.randn(3, args.height, args.width, device=args.device, generator=gen),
               ImageAnnotation(f"synthetic_{i}", objs, img_size=(args.width, args.height)))
