"""`Decoder`: head output -> grouped objects.  Mirrors src/sdnet/data/decoders.py:17-179.

The device stage (clamped sigmoid, 5x5 NMS, top-k, offset / embedding gather, masking, anchor x
part association; decoders.py:41-100) is two HIP launches behind `sd_decode`; its packed result
comes back in ONE device-to-host copy and the `ImageAnnotation` assembly (decoders.py:103-159)
runs on that host copy instead of ~200 blocking `.item()` reads per image.
"""
from __future__ import annotations

import numpy as np
import torch

from .. import _lib as L
from ..utils.ops import clamped_sigmoid
from ..utils.types import ImageAnnotation, Keypoint, Object


class Decoder:
    def __init__(self, args):
        self.label_map = args._r_labels
        self.part_map = args._r_parts
        self.anchor_name = args.anchor_name
        self.args = args
        self.down_ratio = args.down_ratio
        self.max_objects = args.max_objects  # K
        self.max_parts = args.max_parts  # P

    # ------------------------------------------------------------------ device stage
    def decode_packed(self, outputs, conf_thresh, dist_thresh, exact_topk=True):
        """Run sd_decode; returns (packed int32 device buffer, (B, K, P, h, w)).
        exact_topk=False drops peaks with score <= conf before the selection: same annotations, fewer candidates."""
        a, a_p, a_sb, a_sc = L.map_view(outputs["anchor_hm"])
        p, p_p, p_sb, p_sc = L.map_view(outputs["part_hm"])
        o, o_p, o_sb, o_sc = L.map_view(outputs["offsets"])
        e, e_p, e_sb, e_sc = L.map_view(outputs["embeddings"])
        L.require_cuda(a, p, o, e)
        B, M, h, w = a.shape
        N = p.shape[1]
        K, P = self.max_objects, self.max_parts
        lib = L.lib()
        packed = torch.empty(lib.sd_decode_packed_words(B, K, P), dtype=torch.int32, device=a.device)
        ws = L.workspace(lib.sd_decode_workspace_bytes(B, M, N, h, w, K, P), a.device)
        conf32 = float(np.float32(conf_thresh))                       # tensor-vs-scalar compares run in fp32
        dist32 = float(np.float32(dist_thresh * min(w, h)))           # decoders.py:100
        L.check(lib.sd_decode(a_p, a_sb, a_sc, p_p, p_sb, p_sc, o_p, o_sb, o_sc, e_p, e_sb, e_sc, B, M, N, h, w, K, P,
                              conf32, dist32, int(bool(exact_topk)), packed.data_ptr(), ws.data_ptr(), ws.numel(), L.stream()), "sd_decode")
        return packed, (B, K, P, h, w)

    @staticmethod
    def split_packed(packed, B, K, P):
        """Views into the packed buffer (layout: include/sdnet_hip.h, sd_decode).  Works for torch and numpy."""
        f = packed.view(torch.float32) if isinstance(packed, torch.Tensor) else packed.view(np.float32)
        sizes = [("anchor_out", B * K * 4, (B, K, 4), True), ("part_out", B * P * 6, (B, P, 6), True),
                 ("part_emb", B * P * 2, (B, P, 2), True), ("anchor_smask", B * K, (B, K), True),
                 ("part_smask", B * P, (B, P), True), ("anchor_ind", B * K, (B, K), False),
                 ("part_ind", B * P, (B, P), False), ("assign", B * P, (B, P), False)]
        out, off = {}, 0
        for name, n, shape, is_f in sizes:
            src = f if is_f else packed
            out[name] = src[off:off + n].reshape(shape)
            off += n
        return out

    # ------------------------------------------------------------------ reference entry point
    def __call__(self, outputs, conf_thresh=None, dist_thresh=None, return_metadata=False):
        conf_thresh = conf_thresh if conf_thresh is not None else self.args.conf_threshold
        dist_thresh = dist_thresh if dist_thresh is not None else self.args.decoder_dist_thresh

        # the metadata exposes every top-k slot (also peaks below the threshold): exact selection only when it is asked for
        packed, (B, K, P, out_h, out_w) = self.decode_packed(outputs, conf_thresh, dist_thresh, exact_topk=return_metadata)
        in_h, in_w = int(self.down_ratio * out_h), int(self.down_ratio * out_w)       # decoders.py:41
        host = self.split_packed(packed.cpu().numpy(), B, K, P)                        # the one D2H (+ sync)
        sx, sy = in_w / out_w, in_h / out_h                                            # utils.py:19-26

        anchor_out = host["anchor_out"].astype(np.float64)        # float(np.float32) == tensor.item()
        part_out = host["part_out"].astype(np.float64)
        assign = host["assign"]
        annotations = []
        for b in range(B):                                                             # decoders.py:104-139
            by_anchor = {}
            for i in np.nonzero(assign[b] >= 0)[0]:
                by_anchor.setdefault(int(assign[b, i]), []).append(i)
            ann = ImageAnnotation(f"batch_{b}")
            for a in np.nonzero(anchor_out[b, :, 2] > conf_thresh)[0]:                 # skip score <= conf
                ax, ay, asc, alab = anchor_out[b, a].tolist()
                parts = [Keypoint(self.part_map[int(part_out[b, i, 3])], part_out[b, i, 0].item() * sx,
                                  part_out[b, i, 1].item() * sy, part_out[b, i, 2].item())
                         for i in by_anchor.get(int(a), ())]
                anchor = Keypoint(self.anchor_name, ax * sx, ay * sy, asc)
                ann.objects.append(Object(name=self.label_map[int(alab)], anchor=anchor, parts=parts))
            annotations.append(ann)

        if not return_metadata:
            return annotations

        raw_parts = []                                                                 # decoders.py:142-159
        for b in range(B):
            keep = np.nonzero(~(part_out[b, :, 2] < conf_thresh))[0]
            raw_parts.append([Keypoint(self.part_map[int(part_out[b, i, 3])], part_out[b, i, 0].item() * sx,
                                       part_out[b, i, 1].item() * sy, part_out[b, i, 2].item()) for i in keep])

        dev = self.split_packed(packed, B, K, P)
        return {
            "annotation": annotations,
            "anchor_hm_sig": clamped_sigmoid(outputs["anchor_hm"]),
            "part_hm_sig": clamped_sigmoid(outputs["part_hm"]),
            "embeddings": dev["part_emb"],
            "topk_anchor": (dev["anchor_smask"], dev["anchor_ind"].long(), dev["anchor_out"][..., 3],
                            dev["anchor_out"][..., 1], dev["anchor_out"][..., 0]),
            "topk_kp": (dev["part_smask"], dev["part_ind"].long(), dev["part_out"][..., 3],
                        dev["part_out"][..., 1], dev["part_out"][..., 0]),
            "raw_parts": raw_parts,
            "raw_embeddings": outputs["embeddings"],
            "raw_offsets": outputs["offsets"],
        }
