"""GPU tensor primitives with the reference's names and signatures
(src/sdnet/utils/utils.py:341-361,418-467), each one a HIP kernel behind the C ABI."""
from __future__ import annotations

import torch

from .. import _lib as L


def clamped_sigmoid(input: torch.Tensor) -> torch.Tensor:
    """utils.py:355-361."""
    L.require_cuda(input)
    x = input.contiguous().float()
    y = torch.empty_like(x)
    L.check(L.lib().sd_clamped_sigmoid(x.data_ptr(), y.data_ptr(), x.numel(), L.stream()), "sd_clamped_sigmoid")
    return y


def clamp_in_0_1(tensor: torch.Tensor) -> torch.Tensor:
    """utils.py:360-361.  API-parity helper only: nothing on the path calls it (the kernels clamp inline)."""
    return torch.clamp(tensor, min=1e-6, max=1 - 1e-6)


def nms(heatmaps: torch.Tensor) -> torch.Tensor:
    """utils.py:441-443: (hm == maxpool5x5(hm)) * hm."""
    L.require_cuda(heatmaps)
    t, p, sb, sc = L.map_view(heatmaps)
    B, C, h, w = t.shape
    out = torch.empty((B, C, h, w), dtype=torch.float32, device=t.device)
    L.check(L.lib().sd_nms5(p, sb, sc, out.data_ptr(), B, C, h, w, 0, L.stream()), "sd_nms5")
    return out


def _peak_outputs(B, k, device):
    return (torch.empty((B, k), dtype=torch.float32, device=device), torch.empty((B, k), dtype=torch.int64, device=device),
            torch.empty((B, k), dtype=torch.float32, device=device), torch.empty((B, k), dtype=torch.float32, device=device),
            torch.empty((B, k), dtype=torch.float32, device=device))


def topk(scores: torch.Tensor, k: int = 100):
    """utils.py:447-467 -> (score, ind int64, cls, ys, xs), each (B, k).  Ties: class asc, index asc."""
    L.require_cuda(scores)
    t, p, sb, sc = L.map_view(scores)
    B, C, h, w = t.shape
    outs = _peak_outputs(B, k, t.device)
    nbytes = L.lib().sd_topk_workspace_bytes(B, C, h, w, k)
    ws = L.workspace(nbytes, t.device)
    L.check(L.lib().sd_topk(p, sb, sc, B, C, h, w, k, *[o.data_ptr() for o in outs], ws.data_ptr(), ws.numel(), L.stream()),
            "sd_topk")
    return outs


def decode_peaks(logits: torch.Tensor, k: int):
    """Fused clamped_sigmoid + nms + topk (decoders.py:44-48) reading the logits once."""
    L.require_cuda(logits)
    t, p, sb, sc = L.map_view(logits)
    B, C, h, w = t.shape
    outs = _peak_outputs(B, k, t.device)
    nbytes = L.lib().sd_decode_peaks_workspace_bytes(B, C, h, w, k)
    ws = L.workspace(nbytes, t.device)
    L.check(L.lib().sd_decode_peaks(p, sb, sc, B, C, h, w, k, *[o.data_ptr() for o in outs], ws.data_ptr(), ws.numel(),
                                    L.stream()), "sd_decode_peaks")
    return outs


def gather(feat: torch.Tensor, ind: torch.Tensor) -> torch.Tensor:
    """utils.py:341-343 on (B, J) / (B, n) operands (same HIP gather as transpose_and_gather, with one channel)."""
    B, J = feat.shape
    return transpose_and_gather(feat.reshape(B, 1, J, 1), ind)[..., 0]


def transpose_and_gather(feat: torch.Tensor, ind: torch.Tensor) -> torch.Tensor:
    """utils.py:347-351: feat (B,C,H,W), ind (B,n) int64 -> (B,n,C)."""
    L.require_cuda(feat, ind)
    t, p, sb, sc = L.map_view(feat)
    B, C, h, w = t.shape
    ind = ind.contiguous().long()
    n = ind.shape[1]
    out = torch.empty((B, n, C), dtype=torch.float32, device=t.device)
    L.check(L.lib().sd_transpose_and_gather(p, sb, sc, B, C, h * w, ind.data_ptr(), n, out.data_ptr(), L.stream()),
            "sd_transpose_and_gather")
    return out


def hypot(input: torch.Tensor, dim: int = -1, *, output=None) -> torch.Tensor:
    """utils.py:422-437: sqrt(sum(square(x), dim)) for a size-2 dim, fp32, no FMA."""
    assert input.size(dim) == 2, f"the size of dimension {dim} ({input.size(dim)}) should be 2"
    L.require_cuda(input)
    x = input.movedim(dim, -1).contiguous().float()
    out = torch.empty(x.shape[:-1], dtype=torch.float32, device=x.device)
    L.check(L.lib().sd_hypot(x.data_ptr(), out.data_ptr(), out.numel(), L.stream()), "sd_hypot")
    return out


def gaussian_2d(X, Y, mu1, mu2, sigma):
    """utils.py:418-419.  API-parity helper only (plain tensor expression): targets are rendered by `Encode`'s HIP kernel."""
    return torch.exp((-((X - mu1) ** 2) - (Y - mu2) ** 2) / (2 * sigma**2))
