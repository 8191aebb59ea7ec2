"""`Loss`: heatmap (MSE or CornerNet focal) + masked-L1 offset / embedding loss.

Mirrors src/sdnet/model/loss.py:8-64,91-165 (`Loss`, `LossStats`; `FocalLoss` / `L1Loss` exist
as the fused HIP kernels `sd_loss_fwd` / `sd_loss_bwd`).  Differences in mechanism, not in
result: one fused forward pass + one finalize block instead of ~40 launches, and the
reference's host branches (`numel == 0`, `num_pos == 0`) run on the device, so a training step
never synchronises with the host.
"""
from __future__ import annotations

import ctypes as C

import torch

from .. import _lib as L

_HM_FN = {"mse": 0, "focal": 1}


def _common_base(views):
    """If the four head views are channel slices of one contiguous (B, C, h, w) tensor
    (network.py:77-84), return that tensor so the backward writes its gradient in one piece."""
    a = views[0]
    base = a._base
    if base is None or base.dim() != 4 or not base.is_contiguous() or base.dtype != torch.float32:
        return None
    B, Cb, h, w = base.shape
    c0 = 0
    for v in views:
        if v._base is not base or v.shape[0] != B or tuple(v.shape[2:]) != (h, w) or v.stride() != base.stride():
            return None
        if v.storage_offset() != base.storage_offset() + c0 * h * w:
            return None
        c0 += v.shape[1]
    return base if c0 == Cb else None


def loss_forward(head, tgt, cfg):
    """sd_loss_fwd on the raw head tensor (B, M+N+4, h, w).  Returns (desc, keep-alive list, out8) where
    out8 = [total, hm, offset, embedding, num_pos_a, num_pos_p, n_valid_a, n_valid_p] on the device."""
    (M, N, K, P, fn, hm_w, off_w, emb_w) = cfg
    B, Cc, h, w = head.shape
    keep = [head]
    used = ("anchor_hm", "part_hm", "anchor_inds", "part_inds", "anchor_offsets", "part_offsets", "embeddings", "anchor_mask", "part_mask")
    L.require_cuda(head, *[tgt[k] for k in used])      # raw pointers go to the kernels: a host tensor here would fault the GPU
    for k, n in (("anchor_inds", K), ("part_inds", P), ("anchor_mask", K), ("part_mask", P)):
        if tuple(tgt[k].shape) != (B, n):
            raise L.SdError(f"target['{k}'] has shape {tuple(tgt[k].shape)}, expected {(B, n)}")
    for k, n in (("anchor_offsets", K), ("part_offsets", P), ("embeddings", P)):
        if tuple(tgt[k].shape) != (B, n, 2):
            raise L.SdError(f"target['{k}'] has shape {tuple(tgt[k].shape)}, expected {(B, n, 2)}")
    if tuple(tgt["anchor_hm"].shape) != (B, M, h, w) or tuple(tgt["part_hm"].shape) != (B, N, h, w) or Cc != M + N + 4:
        raise L.SdError("heatmap targets / head channels do not match the label and part counts")

    def mp(t):
        t, p, sb, sc = L.map_view(t)
        keep.append(t)
        return p, sb, sc

    def flat(t, dtype):
        if t.dtype == torch.bool and dtype == torch.uint8:
            t = t.view(torch.uint8)
        t = t.to(dtype).contiguous() if t.dtype != dtype or not t.is_contiguous() else t
        keep.append(t)
        return t.data_ptr()

    d = L.LossDesc()
    d.anchor_hm, d.a_sb, d.a_sc = mp(head[:, :M])
    d.part_hm, d.p_sb, d.p_sc = mp(head[:, M:M + N])
    d.offsets, d.o_sb, d.o_sc = mp(head[:, M + N:M + N + 2])
    d.embeddings, d.e_sb, d.e_sc = mp(head[:, M + N + 2:M + N + 4])
    d.t_anchor_hm, d.ta_sb, d.ta_sc = mp(tgt["anchor_hm"])
    d.t_part_hm, d.tp_sb, d.tp_sc = mp(tgt["part_hm"])
    d.anchor_inds = flat(tgt["anchor_inds"], torch.int64)
    d.part_inds = flat(tgt["part_inds"], torch.int64)
    d.anchor_offsets = flat(tgt["anchor_offsets"], torch.float32)
    d.part_offsets = flat(tgt["part_offsets"], torch.float32)
    d.t_embeddings = flat(tgt["embeddings"], torch.float32)
    d.anchor_mask = flat(tgt["anchor_mask"], torch.uint8)
    d.part_mask = flat(tgt["part_mask"], torch.uint8)
    d.B, d.M, d.N, d.h, d.w, d.K, d.P = B, M, N, h, w, K, P
    d.hm_loss_fn = fn
    d.hm_weight, d.offset_weight, d.embedding_weight = hm_w, off_w, emb_w
    out8 = torch.empty(8, dtype=torch.float32, device=head.device)
    lib = L.lib()
    ws = L.workspace(lib.sd_loss_workspace_bytes(B, M, N, h, w), head.device)
    L.check(lib.sd_loss_fwd(C.byref(d), out8.data_ptr(), ws.data_ptr(), ws.numel(), L.stream()), "sd_loss_fwd")
    return d, keep, out8


def loss_backward(desc, out8, grad_out, shape):
    """sd_loss_bwd: d total / d head, (B, M+N+4, h, w) contiguous.  grad_out: 0-dim device tensor."""
    dhead = torch.empty(shape, dtype=torch.float32, device=out8.device)
    g = grad_out.to(torch.float32).contiguous()
    L.check(L.lib().sd_loss_bwd(C.byref(desc), out8.data_ptr(), g.data_ptr(), dhead.data_ptr(), L.stream()), "sd_loss_bwd")
    return dhead


def loss_config(args, M, N, K, P):
    return (M, N, K, P, _HM_FN[args.hm_loss_fn.lower()], float(args.hm_weight), float(args.offset_weight),
            float(args.embedding_weight))


class _SdLoss(torch.autograd.Function):
    @staticmethod
    def forward(ctx, head, tgt, cfg):
        d, keep, out8 = loss_forward(head, tgt, cfg)
        ctx.desc, ctx.keep, ctx.out8, ctx.shape = d, keep, out8, tuple(head.shape)
        ctx.mark_non_differentiable(out8)
        return out8[0].clone(), out8

    @staticmethod
    def backward(ctx, g_total, _g_out8):
        return loss_backward(ctx.desc, ctx.out8, g_total, ctx.shape), None, None


class Loss(torch.nn.Module):
    def __init__(self, args):
        super().__init__()
        self.args = args
        if args.hm_loss_fn.lower() not in _HM_FN:
            raise IOError(f"'hm_loss_fn' should either be 'focal' or 'mse', not {args.hm_loss_fn}.")
        self.stats = LossStats()

    def forward(self, input, target):
        views = [input["anchor_hm"], input["part_hm"], input["offsets"], input["embeddings"]]
        L.require_cuda(*views)
        M, N = views[0].shape[1], views[1].shape[1]
        head = _common_base(views)
        if head is None:                      # separate tensors: one concat, autograd splits the gradient back
            head = torch.cat([v.float() for v in views], dim=1)
        cfg = loss_config(self.args, M, N, target["anchor_inds"].shape[1], target["part_inds"].shape[1])
        total, out8 = _SdLoss.apply(head, target, cfg)
        self.stats.update(out8[1], out8[2], out8[3])
        return total


class LossStats:
    """Three-field accumulator, src/sdnet/model/loss.py:120-165."""

    def __init__(self, hm_loss=0.0, offset_loss=0.0, embedding_loss=0.0):
        self.hm_loss, self.offset_loss, self.embedding_loss = hm_loss, offset_loss, embedding_loss

    def reset(self):
        self.hm_loss = self.offset_loss = self.embedding_loss = 0.0

    def update(self, hm_loss, offset_loss, embedding_loss):
        self.hm_loss, self.offset_loss, self.embedding_loss = hm_loss, offset_loss, embedding_loss

    @property
    def total_loss(self):
        return self.hm_loss + self.offset_loss + self.embedding_loss

    def _zip(self, other, op):
        return [op(getattr(self, k), getattr(other, k) if isinstance(other, LossStats) else other)
                for k in ("hm_loss", "offset_loss", "embedding_loss")]

    def __add__(self, other):
        return LossStats(*self._zip(other, lambda a, b: a + b))

    def __iadd__(self, other):
        self.update(*self._zip(other, lambda a, b: a + b))
        return self

    def __truediv__(self, value):
        return LossStats(*self._zip(value, lambda a, b: a / b))

    def __itruediv__(self, value):
        self.update(*self._zip(value, lambda a, b: a / b))
        return self

    def __repr__(self):
        return (f"total_loss: {self.total_loss}, hm_loss: {self.hm_loss}, offset_loss: {self.offset_loss}, "
                f"embedding_loss: {self.embedding_loss}")
