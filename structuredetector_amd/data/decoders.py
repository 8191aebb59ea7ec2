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
        self._state = {}
        self._plans = {}
        self._thresholds = {}
        self.selector_timeouts = 0       # sd_decode_fused calls that were redone by the two-launch decoder (a tile block was delayed)

    # ------------------------------------------------------------------ device stage
    def decode_packed(self, outputs, conf_thresh, dist_thresh, exact_topk=True, fused=None, state_out=None):
        """Run the device stage; returns (packed int32 device buffer, (B, K, P, h, w)).
        exact_topk=False drops peaks with score < fp32(conf) before the selection: same annotations, fewer candidates.
        fused: None = one-launch sd_decode_fused whenever K, P allow it; False = the two-launch sd_decode (bit-identical).
        state_out: a list; when given and the one-launch kernel runs, it runs on a hand-off state buffer of ITS OWN taken from this
        decoder's free list and appended to the list (the caller gives it back with `_release_state`): a decode whose status has not
        been read yet must not share its records with the next launch (`submit`)."""
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
        conf32 = float(np.float32(conf_thresh))                       # tensor-vs-scalar compares run in fp32
        dist32 = float(np.float32(dist_thresh * min(w, h)))           # decoders.py:100
        force = 2 if fused else 0            # an explicit fused=True runs the one-launch kernel also where the launch pair is faster (tests)
        if fused is None:
            # the measured rule lives behind the C ABI (sd_decode_fused_recommended): one launch for image geometries up to 256 tile
            # blocks, and for the exact top-k only up to batch 8
            fused = bool(lib.sd_decode_fused_recommended(B, M, N, h, w, K, P, int(exact_topk)))
        if fused:
            ws = L.workspace(lib.sd_decode_fused_workspace_bytes(B, M, N, h, w, K, P), a.device)
            need = lib.sd_decode_state_bytes(B, M, N, h, w)
            if state_out is None:
                state = self._fused_state(a.device, need)
            else:
                state = self._acquire_state(a.device, need)
                state_out.append(state)
            try:
                L.check(lib.sd_decode_fused(a_p, a_sb, a_sc, p_p, p_sb, p_sc, o_p, o_sb, o_sc, e_p, e_sb, e_sc, B, M, N, h, w, K, P,
                                            conf32, dist32, int(bool(exact_topk)) | force, packed.data_ptr(), state.data_ptr(), state.numel(),
                                            ws.data_ptr(), ws.numel(), L.stream()), "sd_decode_fused")
            except L.SdError:
                state.zero_()                                         # contract: re-zero the state after a failed call
                raise
        else:
            ws = L.workspace(lib.sd_decode_workspace_bytes(B, M, N, h, w, K, P), a.device)
            L.check(lib.sd_decode(a_p, a_sb, a_sc, p_p, p_sb, p_sc, o_p, o_sb, o_sc, e_p, e_sb, e_sc, B, M, N, h, w, K, P,
                                  conf32, dist32, int(bool(exact_topk)), packed.data_ptr(), ws.data_ptr(), ws.numel(), L.stream()), "sd_decode")
        return packed, (B, K, P, h, w)

    def _fused_state(self, device, need):
        """Hand-off records of sd_decode_fused: zero once, left zero by every call; one buffer per (device, stream)."""
        key = (device.index if device.index is not None else torch.cuda.current_device(), L.stream())
        buf = self._state.get(key)
        if buf is None or buf.numel() < need:
            buf = torch.zeros(max(need, 1 << 14), dtype=torch.uint8, device=device)
            self._state[key] = buf
        return buf

    def _acquire_state(self, device, need):
        """A zeroed hand-off state buffer owned by ONE in-flight `submit` (ring of as many buffers as there are submissions in flight):
        sd_decode_fused leaves its state zero only when every selector met its tiles; a timed-out call leaves a late tile's record
        behind, and that record is self-validating -- the NEXT launch on the same buffer would accept it.  The synchronous `__call__`
        reads the status and re-zeroes before its next launch; a queued submission cannot, so it gets its own buffer."""
        key = (device.index if device.index is not None else torch.cuda.current_device(), L.stream())
        free = self.__dict__.setdefault("_state_free", {}).setdefault(key, [])
        for i, buf in enumerate(free):
            if buf.numel() >= need:
                return free.pop(i)
        return torch.zeros(max(need, 1 << 14), dtype=torch.uint8, device=device)

    def _release_state(self, state, key, failed):
        """Back to the free list; after a failed call (caller has synchronised the stream) the buffer is re-zeroed first."""
        if failed:
            state.zero_()
        self.__dict__.setdefault("_state_free", {}).setdefault(key, []).append(state)

    @staticmethod
    def split_packed(packed, B, K, P):
        """Views into the packed buffer (layout: include/sdnet_hip.h, sd_decode).  Works for torch and numpy."""
        f = packed.view(torch.float32) if isinstance(packed, torch.Tensor) else packed.view(np.float32)
        sizes = [("anchor_out", B * K * 4, (B, K, 4), True), ("part_out", B * P * 6, (B, P, 6), True),
                 ("part_emb", B * P * 2, (B, P, 2), True), ("anchor_smask", B * K, (B, K), True),
                 ("part_smask", B * P, (B, P), True), ("anchor_ind", B * K, (B, K), False),
                 ("part_ind", B * P, (B, P), False), ("assign", B * P, (B, P), False), ("status", B, (B,), False)]
        out, off = {}, 0
        for name, n, shape, is_f in sizes:
            src = f if is_f else packed
            out[name] = src[off:off + n].reshape(shape)
            off += n
        return out

    # ------------------------------------------------------------------ reference entry point
    def _call_low_latency(self, outputs, conf_thresh, dist_thresh):
        """Decoder.__call__ without metadata on the one-launch kernel, trimmed for bs = 1 latency: the kernel writes its packed
        result straight into pinned (device-mapped) host memory -- no device buffer, no copy launch -- the host waits on the
        stream once and assembles the objects from plain Python lists.  Same values as the general path (tests assert it).
        Returns None when the geometry is not served by sd_decode_fused (caller falls back)."""
        a, a_p, a_sb, a_sc = L.map_view(outputs["anchor_hm"])
        p, p_p, p_sb, p_sc = L.map_view(outputs["part_hm"])
        o, o_p, o_sb, o_sc = L.map_view(outputs["offsets"])
        e, e_p, e_sb, e_sc = L.map_view(outputs["embeddings"])
        if not (a.is_cuda and p.is_cuda and o.is_cuda and e.is_cuda):
            L.require_cuda(a, p, o, e)
        B, M, h, w = a.shape
        N = p.shape[1]
        K, P = self.max_objects, self.max_parts
        stream = L.stream()
        key = (a.device.index, stream, B, M, N, h, w)           # per stream: scratch and hand-off state must not be shared in flight
        plan = self._plans.get(key)
        lib = L.lib()
        if plan is None:
            if not lib.sd_decode_fused_recommended(B, M, N, h, w, K, P, 0):
                self._plans[key] = False
                return None
            host = torch.empty(lib.sd_decode_packed_words(B, K, P), dtype=torch.int32, pin_memory=True)
            # scratch and hand-off state owned by the plan (the shared grow-only workspace may be re-allocated by other ops)
            ws = torch.empty(lib.sd_decode_fused_workspace_bytes(B, M, N, h, w, K, P), dtype=torch.uint8, device=a.device)
            state = torch.zeros(lib.sd_decode_state_bytes(B, M, N, h, w), dtype=torch.uint8, device=a.device)
            views = self.split_packed(host.numpy(), B, K, P)
            plan = (host.data_ptr(), views["anchor_out"], views["part_out"], views["assign"], views["status"], ws.data_ptr(), ws.numel(),
                    state.data_ptr(), state.numel(), (host, ws, state))
            self._plans[key] = plan
        elif plan is False:
            return None
        host_ptr, v_anchor, v_part, v_assign, v_status, ws_ptr, ws_n, st_ptr, st_n, keep = plan
        th = self._thresholds.get((conf_thresh, dist_thresh, w, h))
        if th is None:
            th = self._thresholds[(conf_thresh, dist_thresh, w, h)] = (float(np.float32(conf_thresh)), float(np.float32(dist_thresh * min(w, h))))
        rc = lib.sd_decode_fused(a_p, a_sb, a_sc, p_p, p_sb, p_sc, o_p, o_sb, o_sc, e_p, e_sb, e_sc, B, M, N, h, w, K, P,
                                 th[0], th[1], 0, host_ptr, st_ptr, st_n, ws_ptr, ws_n, stream)
        if rc == 0:
            rc = lib.sd_stream_synchronize(stream)                                     # the one host wait
        status = v_status.tolist()
        if rc or any(status):
            # a selector gave up its bounded wait for a tile block (e.g. another process or stream held the GPU): re-zero THIS call's
            # hand-off state -- the stream is idle, nothing else uses the buffer -- and let the caller take the two-launch path once
            keep[2].zero_()
            L.check(rc, "sd_decode_fused")
            self.selector_timeouts += 1
            return None
        in_h, in_w = int(self.down_ratio * h), int(self.down_ratio * w)                # decoders.py:41
        sx, sy = in_w / w, in_h / h                                                    # utils.py:19-26
        anchor_all, part_all, assign_all = v_anchor.tolist(), v_part.tolist(), v_assign.tolist()
        label_map, part_map, anchor_name = self.label_map, self.part_map, self.anchor_name
        annotations = []
        for b in range(B):                                                             # decoders.py:104-139
            parts_b = part_all[b]
            by_anchor = {}
            for i, an in enumerate(assign_all[b]):
                if an >= 0:
                    by_anchor.setdefault(an, []).append(i)
            ann = ImageAnnotation(f"batch_{b}")
            objs = ann.objects
            for an, (ax, ay, asc, alab) in enumerate(anchor_all[b]):
                if asc > conf_thresh:                                                  # skip score <= conf (double compare)
                    parts = [Keypoint(part_map[int(parts_b[i][3])], parts_b[i][0] * sx, parts_b[i][1] * sy, parts_b[i][2])
                             for i in by_anchor.get(an, ())]
                    objs.append(Object(name=label_map[int(alab)], anchor=Keypoint(anchor_name, ax * sx, ay * sy, asc), parts=parts))
            annotations.append(ann)
        return annotations

    def _redo_two_launch(self, outputs, conf_thresh, dist_thresh, exact_topk, device):
        """A selector of sd_decode_fused gave up its bounded wait for a tile block (transient: another process or stream held the GPU).
        The caller has synchronised this stream, so ITS state buffer -- and only its -- can be re-zeroed; the two-launch decoder has
        no cross-block wait: retry once through it."""
        self.selector_timeouts += 1
        self._fused_state(device, 0).zero_()
        packed, (B, K, P, _, _) = self.decode_packed(outputs, conf_thresh, dist_thresh, exact_topk=exact_topk, fused=False)
        host = self.split_packed(self._to_host(packed), B, K, P)
        if host["status"].any():
            raise L.SdError(f"sd_decode: non-zero status for image(s) {np.nonzero(host['status'])[0].tolist()}")
        return packed, host

    def _pinned(self, n, dtype):
        """A pinned host buffer of n elements from this decoder's free list (returned by `PendingDecode.result`)."""
        pool = self.__dict__.setdefault("_host_pool", {})
        free = pool.setdefault((n, dtype), [])
        return free.pop() if free else torch.empty(n, dtype=dtype, pin_memory=True)

    def _to_host(self, packed):
        """The packed result on the host: an asynchronous copy into a pinned buffer of this decoder + one stream wait.  (`tensor.cpu()`
        into pageable memory took 0.6 ms of a 0.72 ms call for these 1.4 KB per image; the values are consumed -- copied into Python
        objects -- before the next call reuses the buffer.)"""
        n = packed.numel()
        buf = self._host_buf.get(n) if hasattr(self, "_host_buf") else None
        if buf is None:
            if not hasattr(self, "_host_buf"):
                self._host_buf = {}
            buf = self._host_buf[n] = torch.empty(n, dtype=packed.dtype, pin_memory=True)
        buf.copy_(packed, non_blocking=True)
        torch.cuda.current_stream(packed.device).synchronize()
        return buf.numpy()

    def _assemble(self, host, B, out_h, out_w, conf_thresh, want_raw):
        """decoders.py:103-159 on the host copy of the packed result: (annotations, raw_parts or None).  Values leave numpy ONCE
        (`tolist()`: float(np.float32) == tensor.item()); the per-image loops run on plain Python lists."""
        in_h, in_w = int(self.down_ratio * out_h), int(self.down_ratio * out_w)       # decoders.py:41
        sx, sy = in_w / out_w, in_h / out_h                                            # utils.py:19-26
        anchor_all, part_all, assign_all = host["anchor_out"].tolist(), host["part_out"].tolist(), host["assign"].tolist()
        label_map, part_map, anchor_name = self.label_map, self.part_map, self.anchor_name
        annotations, raw_parts = [], ([] if want_raw else None)
        for b in range(B):                                                             # decoders.py:104-139
            parts_b = part_all[b]
            by_anchor = {}
            for i, an in enumerate(assign_all[b]):
                if an >= 0:
                    by_anchor.setdefault(an, []).append(i)
            ann = ImageAnnotation(f"batch_{b}")
            objs = ann.objects
            for an, (ax, ay, asc, alab) in enumerate(anchor_all[b]):
                if asc > conf_thresh:                                                  # skip score <= conf (double compare)
                    parts = [Keypoint(part_map[int(parts_b[i][3])], parts_b[i][0] * sx, parts_b[i][1] * sy, parts_b[i][2])
                             for i in by_anchor.get(an, ())]
                    objs.append(Object(name=label_map[int(alab)], anchor=Keypoint(anchor_name, ax * sx, ay * sy, asc), parts=parts))
            annotations.append(ann)
            if want_raw:                                                               # decoders.py:142-159: skip score < conf
                raw_parts.append([Keypoint(part_map[int(pt[3])], pt[0] * sx, pt[1] * sy, pt[2]) for pt in parts_b if not pt[2] < conf_thresh])
        return annotations, raw_parts

    def submit(self, outputs, conf_thresh=None, dist_thresh=None, with_raw_parts=False):
        """The device stage of `__call__` WITHOUT the host wait: launches the decoder, starts the copy of the packed result into a pinned
        buffer and returns a `PendingDecode`; `.result()` waits for that copy and assembles `(annotations, raw_parts | None)`.  Lets a
        batched caller (`evaluate`, validation, `detect`) queue the next batch's forward + decode before it assembles this one."""
        conf_thresh = conf_thresh if conf_thresh is not None else self.args.conf_threshold
        dist_thresh = dist_thresh if dist_thresh is not None else self.args.decoder_dist_thresh
        states = []
        packed, (B, K, P, out_h, out_w) = self.decode_packed(outputs, conf_thresh, dist_thresh, exact_topk=with_raw_parts, state_out=states)
        buf = self._pinned(packed.numel(), packed.dtype)
        buf.copy_(packed, non_blocking=True)
        done = torch.cuda.Event()
        done.record(torch.cuda.current_stream(packed.device))
        dev = packed.device
        state_key = (dev.index if dev.index is not None else torch.cuda.current_device(), L.stream())
        return PendingDecode(self, outputs, packed, buf, done, (B, K, P, out_h, out_w), conf_thresh, dist_thresh, with_raw_parts,
                             states[0] if states else None, state_key)

    def __call__(self, outputs, conf_thresh=None, dist_thresh=None, return_metadata=False, metadata_fields=None):
        """decoders.py:29-179.  `metadata_fields` (extension, default None = the reference's full metadata dict): the keys a caller of
        `return_metadata=True` will read -- `evaluate` and the validation pass only use "annotation" and "raw_parts", and the other
        entries cost two sigmoid launches over the heatmaps and a dozen device tensor views per image."""
        conf_thresh = conf_thresh if conf_thresh is not None else self.args.conf_threshold
        dist_thresh = dist_thresh if dist_thresh is not None else self.args.decoder_dist_thresh

        if not return_metadata:
            fast = self._call_low_latency(outputs, conf_thresh, dist_thresh)
            if fast is not None:
                return fast

        # the metadata exposes every top-k slot (also peaks below the threshold): exact selection only when it is asked for
        packed, (B, K, P, out_h, out_w) = self.decode_packed(outputs, conf_thresh, dist_thresh, exact_topk=return_metadata)
        host = self.split_packed(self._to_host(packed), B, K, P)                       # the one D2H (+ sync)
        if host["status"].any():
            packed, host = self._redo_two_launch(outputs, conf_thresh, dist_thresh, return_metadata, packed.device)
        annotations, raw_parts = self._assemble(host, B, out_h, out_w, conf_thresh, return_metadata)
        if not return_metadata:
            return annotations

        if metadata_fields is not None and set(metadata_fields) <= {"annotation", "raw_parts"}:
            return {"annotation": annotations, "raw_parts": raw_parts}
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


class PendingDecode:
    """A decode in flight (`Decoder.submit`): device buffer, pinned host copy, the event after which the copy is complete."""

    def __init__(self, decoder, outputs, packed, host, done, dims, conf_thresh, dist_thresh, want_raw, state=None, state_key=None):
        self.decoder, self.outputs, self.packed, self.host, self.done = decoder, outputs, packed, host, done
        self.dims, self.conf_thresh, self.dist_thresh, self.want_raw = dims, conf_thresh, dist_thresh, want_raw
        self.state, self.state_key = state, state_key      # the one-launch kernel's hand-off records: this submission's own buffer

    def result(self):
        dec = self.decoder
        B, K, P, out_h, out_w = self.dims
        self.done.synchronize()
        host = dec.split_packed(self.host.numpy(), B, K, P)
        failed = bool(host["status"].any())
        if failed:
            torch.cuda.current_stream(self.packed.device).synchronize()
        if self.state is not None:
            dec._release_state(self.state, self.state_key, failed)       # a late tile's record may still sit in it: re-zeroed
            self.state = None
        if failed:
            _, host = dec._redo_two_launch(self.outputs, self.conf_thresh, self.dist_thresh, self.want_raw, self.packed.device)
        out = dec._assemble(host, B, out_h, out_w, self.conf_thresh, self.want_raw)
        dec._host_pool[(self.host.numel(), self.host.dtype)].append(self.host)        # values were copied into Python objects
        self.outputs = self.packed = self.host = None
        return out


class RawDecoder:
    """src/sdnet/cli/convert_coreml.py:12-18: the part of the decoder that the reference bakes into its exported model --
    `cat(nms(clamped_sigmoid(x[:, :nb_hms])), x[:, nb_hms:])`.  Here ONE tile kernel pass (`sd_nms5` with the sigmoid
    fused: the logits are read once) writes the heatmap channels of the output; the regression channels are copied."""

    def __init__(self, nb_hms: int) -> None:
        self.nb_hms = nb_hms

    def __call__(self, input: torch.Tensor) -> torch.Tensor:
        L.require_cuda(input)
        x = input if input.dtype == torch.float32 else input.float()
        B, C, h, w = x.shape
        if not 0 < self.nb_hms <= C:
            raise L.SdError(f"RawDecoder: nb_hms={self.nb_hms} does not fit a {C}-channel output")
        out = torch.empty((B, C, h, w), dtype=torch.float32, device=x.device)
        t, p, sb, sc = L.map_view(x[:, :self.nb_hms])
        # the NMS output plane (b, c) lives at out[b, c]: channel stride h*w, batch stride C*h*w -> write through a dense scratch
        hm = out[:, :self.nb_hms] if B == 1 else torch.empty((B, self.nb_hms, h, w), dtype=torch.float32, device=x.device)
        L.check(L.lib().sd_nms5(p, sb, sc, hm.data_ptr(), B, self.nb_hms, h, w, 1, L.stream()), "sd_nms5")
        if B != 1:
            out[:, :self.nb_hms] = hm
        out[:, self.nb_hms:] = x[:, self.nb_hms:]
        return out


class FusedOutputDecoder(Decoder):
    """Decoder for the output of a model with the sigmoid + NMS embedded (reference: `CoreMLDecoder`,
    src/sdnet/data/decoders.py:182-342): top-k directly on the already suppressed heatmaps (`sd_topk`), then the same gather +
    association kernel (`sd_decode_group`) and the same host assembly as `Decoder`."""

    def decode_packed(self, outputs, conf_thresh, dist_thresh, exact_topk=True, fused=None, state_out=None):
        from ..utils.ops import topk
        a_s, a_i, a_c, _, _ = topk(outputs["anchor_hm"], self.max_objects)
        p_s, p_i, p_c, _, _ = topk(outputs["part_hm"], self.max_parts)
        o, o_p, o_sb, o_sc = L.map_view(outputs["offsets"])
        e, e_p, e_sb, e_sc = L.map_view(outputs["embeddings"])
        B, _, h, w = o.shape
        K, P = self.max_objects, self.max_parts
        lib = L.lib()
        packed = torch.empty(lib.sd_decode_packed_words(B, K, P), dtype=torch.int32, device=o.device)
        L.check(lib.sd_decode_group(a_s.data_ptr(), a_i.data_ptr(), a_c.data_ptr(), p_s.data_ptr(), p_i.data_ptr(), p_c.data_ptr(),
                                    o_p, o_sb, o_sc, e_p, e_sb, e_sc, B, h, w, K, P, float(np.float32(conf_thresh)),
                                    float(np.float32(dist_thresh * min(w, h))), packed.data_ptr(), L.stream()), "sd_decode_group")
        return packed, (B, K, P, h, w)

    def _call_low_latency(self, outputs, conf_thresh, dist_thresh):
        return None                                   # (the one-launch kernel starts from logits; this class starts from NMS'ed maps)
