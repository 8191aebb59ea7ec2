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
