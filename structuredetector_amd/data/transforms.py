"""`Encode`: annotations -> dense training targets, rendered on the GPU for a whole batch.

Mirrors Encode.__call__, src/sdnet/data/transforms.py:121-208 (keys, shapes and dtypes of the
returned dict, the int() truncation of centres, the double-precision clip + resize of
utils.py:19-26,364-381, and the max_objects / max_parts truncation quirk of :157,186-191).
The reference renders inside DataLoader workers (one full-frame exp per keypoint, 11 ms/img);
here the small index/offset arrays are computed vectorised on the host in float64 and the
heatmaps by one HIP launch per batch (`sd_render_targets`).
"""
from __future__ import annotations

import numpy as np
import torch

from .. import _lib as L
from ..utils import trace as T
from ..utils.misc import clip_annotation


def scenes_to_flat(annotations, labels, parts):
    """list[ImageAnnotation] -> flat float64/int arrays (one Python pass over the objects)."""
    n_obj, o_lab, o_xy, o_np, p_kind, p_xy = [], [], [], [], [], []
    for ann in annotations:
        n_obj.append(len(ann.objects))
        for obj in ann.objects:
            o_lab.append(labels[obj.name]); o_xy.append((obj.x, obj.y)); o_np.append(len(obj.parts))
            for kp in obj.parts:
                p_kind.append(parts[kp.kind]); p_xy.append((kp.x, kp.y))
    return (np.asarray(n_obj, np.int64), np.asarray(o_lab, np.int64), np.asarray(o_xy, np.float64).reshape(-1, 2),
            np.asarray(o_np, np.int64), np.asarray(p_kind, np.int64), np.asarray(p_xy, np.float64).reshape(-1, 2))


class Encode:
    def __init__(self, args):
        self.down_ratio = args.down_ratio
        self.labels = args.labels
        self.parts = args.parts
        self.max_objects = args.max_objects
        self.max_parts = args.max_parts
        self.sigma_gauss = args.sigma_gauss
        self.device = getattr(args, "device", None)

    # ------------------------------------------------------------------ host stage (float64, vectorised)
    def plan(self, img_w, img_h, n_obj, o_lab, o_xy, o_np, p_kind, p_xy):
        """Slot assignment + small target arrays for a batch given as flat arrays.
        n_obj (B) objects per image; o_* per object; o_np parts per object; p_* per part (object-major)."""
        M, K, P = len(self.labels), self.max_objects, self.max_parts
        B = len(n_obj)
        out_w, out_h = int(img_w / self.down_ratio), int(img_h / self.down_ratio)        # transforms.py:138
        rw, rh = out_w / img_w, out_h / img_h                                            # utils.py:23-24
        n_objs = int(n_obj.sum())
        img_of_obj = np.repeat(np.arange(B), n_obj)
        obj_start = np.concatenate(([0], np.cumsum(n_obj)))[:-1]
        rank = np.arange(n_objs) - obj_start[img_of_obj]                                 # object index inside its image
        # parts seen before each object inside its image (untruncated): the loops stop once it reaches P
        cs = np.concatenate(([0], np.cumsum(o_np)))
        before = cs[:-1] - cs[obj_start][img_of_obj] if n_objs else np.zeros(0, np.int64)
        obj_ok = (rank < K) & (before < P)                                               # :157 and :186-191
        ox = np.clip(o_xy[:, 0], 0, img_w - 1) * rw                                      # utils.py:368-369, 23
        oy = np.clip(o_xy[:, 1], 0, img_h - 1) * rh
        obj_of_part = np.repeat(np.arange(n_objs), o_np)
        j = np.arange(int(o_np.sum())) - cs[:-1][obj_of_part] if n_objs else np.zeros(0, np.int64)
        slot = before[obj_of_part] + j                                                   # kp_idx of the part
        part_ok = obj_ok[obj_of_part] & (slot < P)
        px = np.clip(p_xy[:, 0], 0, img_w - 1) * rw
        py = np.clip(p_xy[:, 1], 0, img_h - 1) * rh

        f32 = np.zeros((B * K * 2 + B * P * 4,), np.float32)
        a_off = f32[:B * K * 2].reshape(B, K, 2)
        p_off = f32[B * K * 2:B * K * 2 + B * P * 2].reshape(B, P, 2)
        emb = f32[B * K * 2 + B * P * 2:].reshape(B, P, 2)
        i64 = np.zeros((B * K + B * P,), np.int64)
        a_ind = i64[:B * K].reshape(B, K); p_ind = i64[B * K:].reshape(B, P)
        u8 = np.zeros((B * K + B * P,), np.uint8)
        a_msk = u8[:B * K].reshape(B, K); p_msk = u8[B * K:].reshape(B, P)

        o = np.nonzero(obj_ok)[0]
        oxi, oyi = ox[o].astype(np.int64), oy[o].astype(np.int64)                        # int() truncation, coords >= 0
        a_ind[img_of_obj[o], rank[o]] = oyi * out_w + oxi                                # :163
        a_off[img_of_obj[o], rank[o], 0] = ox[o] - oxi                                   # :165
        a_off[img_of_obj[o], rank[o], 1] = oy[o] - oyi
        a_msk[img_of_obj[o], rank[o]] = 1
        q = np.nonzero(part_ok)[0]
        pb, ps, po = img_of_obj[obj_of_part[q]], slot[q], obj_of_part[q]
        pxi, pyi = px[q].astype(np.int64), py[q].astype(np.int64)
        p_ind[pb, ps] = pyi * out_w + pxi                                                # :176
        p_off[pb, ps, 0] = px[q] - pxi; p_off[pb, ps, 1] = py[q] - pyi                   # :178
        emb[pb, ps, 0] = ox[po] - px[q]; emb[pb, ps, 1] = oy[po] - py[q]                 # :181
        p_msk[pb, ps] = 1

        # keypoint centres for the renderer, sorted by (image, channel)
        C = M + len(self.parts)
        kb = np.concatenate((img_of_obj[o], pb)); kc = np.concatenate((o_lab[o], p_kind[q] + M))
        kx = np.concatenate((oxi, pxi)); ky = np.concatenate((oyi, pyi))
        order = np.lexsort((kc, kb))
        row = (kb * C + kc)[order]
        i32 = np.zeros((2 * len(order) + B * C + 1,), np.int32)
        n = len(order)
        i32[:n] = kx[order]; i32[n:2 * n] = ky[order]
        i32[2 * n:] = np.searchsorted(row, np.arange(B * C + 1), side="left")
        sigma = self.sigma_gauss * min(out_w, out_h) / 3                                 # :142
        return dict(B=B, C=C, M=M, K=K, P=P, out_w=out_w, out_h=out_h, n_kp=n, f32=f32, i64=i64, u8=u8, i32=i32,
                    two_sigma2=float(np.float32(2 * sigma ** 2)))

    # ------------------------------------------------------------------ device stage
    def upload(self, plan, device=None):
        """Host -> device copy of a plan's small arrays (4 transfers per batch)."""
        device = torch.device(device or self.device or "cuda")
        if device.type != "cuda":
            raise L.SdError("Encode renders on the GPU; there is no CPU path")
        dev = {k: plan[k] for k in ("B", "C", "M", "K", "P", "out_h", "out_w", "n_kp", "two_sigma2")}
        for k in ("f32", "i64", "u8", "i32"):
            dev[k] = torch.from_numpy(plan[k]).to(device, non_blocking=True)
        return dev

    def render_device(self, dev):
        """One `sd_render_targets` launch; everything else is views of the uploaded arrays."""
        B, C, M, K, P, h, w, n = (dev[k] for k in ("B", "C", "M", "K", "P", "out_h", "out_w", "n_kp"))
        f32, i64, u8, i32 = dev["f32"], dev["i64"], dev["u8"], dev["i32"]
        hm = torch.empty((B, C, h, w), dtype=torch.float32, device=f32.device)
        base = i32.data_ptr()
        L.check(L.lib().sd_render_targets(base, base + 4 * n, base + 8 * n, B, C, h, w, dev["two_sigma2"], hm.data_ptr(),
                                          L.stream()), "sd_render_targets")
        return {
            "anchor_hm": hm[:, :M], "part_hm": hm[:, M:],
            "anchor_inds": i64[:B * K].view(B, K), "part_inds": i64[B * K:].view(B, P),
            "anchor_offsets": f32[:B * K * 2].view(B, K, 2),
            "part_offsets": f32[B * K * 2:B * K * 2 + B * P * 2].view(B, P, 2),
            "embeddings": f32[B * K * 2 + B * P * 2:].view(B, P, 2),
            "anchor_mask": u8[:B * K].view(B, K).view(torch.bool), "part_mask": u8[B * K:].view(B, P).view(torch.bool),
        }

    def render(self, plan, device=None):
        with T.span("render_targets"):
            return self.render_device(self.upload(plan, device))

    def batch(self, img_size, annotations, device=None):
        """Collated targets (the dict CropDataset.collate_fn would build, dataset.py:58-87) for a list of
        annotations sharing one image size.  Clips the annotations in place like the reference (:154)."""
        img_w, img_h = img_size
        for ann in annotations:
            clip_annotation(ann, (img_w, img_h))
        out = self.render(self.plan(img_w, img_h, *scenes_to_flat(annotations, self.labels, self.parts)), device)
        out["annotation"] = list(annotations)
        return out

    def __call__(self, input, target):
        """Single-sample form with the reference's signature and return layout (no batch dim)."""
        if isinstance(input, torch.Tensor):
            img_h, img_w = input.shape[-2:]
        elif hasattr(input, "size") and not callable(input.size):      # PIL image
            img_w, img_h = input.size
        else:
            raise ValueError(f"`input` type '{type(input)}' not supported")
        dev = input.device if isinstance(input, torch.Tensor) and input.is_cuda else None
        out = self.batch((img_w, img_h), [target], dev)
        res = {k: (v[0] if isinstance(v, torch.Tensor) else v) for k, v in out.items() if k != "annotation"}
        res["image"] = input
        res["annotation"] = target
        return res

    def __repr__(self):
        return (f"Encode(max_objects: {self.max_objects}, max_parts: {self.max_parts}, down_ratio: {self.down_ratio}, "
                f"nb_labels: {len(self.labels)}, nb_parts: {len(self.parts)})")
