"""GPU input pipeline: the reference's per-sample PIL / torchvision chain
`Resize -> [ColorJitter] -> RandomHorizontalFlip -> RandomVerticalFlip -> Normalize` (src/sdnet/data/transforms.py:217-234,
validation :255-261) for a whole batch in two HIP launches (`sd_preprocess_images`), with the annotation transforms of
src/sdnet/utils/utils.py:384-415 on the host and the per-epoch multi-scale shapes of transforms.py:237-244.

Resize parity: `F.resize` of a PIL image is Pillow's separable 8-bit fixed-point resampling; `pil_bilinear_coeffs` reproduces
Pillow's coefficient tables (src/libImaging/Resample.c: precompute_coeffs, normalize_coeffs_8bpc) and the kernels its integer
arithmetic, so the resized bytes equal `Image.resize(size, BILINEAR)` and the normalised tensor equals the reference's bit for bit.
ColorJitter parity (transforms.py:37-47): torchvision's ColorJitter on a PIL image is Pillow's ImageEnhance.Brightness / Contrast /
Color plus an HSV round trip for the hue, in a random order; `sd_preprocess_images_jitter` reproduces Pillow's byte arithmetic on the
resized image (oracle/pil_photometric.py is the restatement, pinned against Pillow over all 2^24 colours; torchvision itself is absent
here, its PIL code path is restated from the published 0.20.1 source), so with the same random draws the normalised tensor equals the
reference's bit for bit.  The random draws come from torch's global generator on the host (TrainAugmentation.draws_for).
"""
from __future__ import annotations

import ctypes as C
import math

import numpy as np
import torch

from .. import _lib as L
from ..utils.misc import clip_annotation, hflip_annotation, vflip_annotation

PRECISION_BITS = 32 - 8 - 2
_MEAN = (0.485, 0.456, 0.406)
_STD = (0.229, 0.224, 0.225)


def pil_bilinear_coeffs(in_size: int, out_size: int):
    """Pillow's precompute_coeffs (triangle filter, support 1.0 scaled by max(in/out, 1)) + normalize_coeffs_8bpc for one axis.
    Returns (bounds int32 (out, 2) = first source index and tap count, kk int32 (out, ksize) 22-bit fixed-point weights, ksize).
    Plain Python floats are C doubles and int() truncates like a C cast, so every intermediate matches Pillow's."""
    scale = filterscale = in_size / out_size
    if filterscale < 1.0:
        filterscale = 1.0
    support = 1.0 * filterscale
    ksize = int(math.ceil(support)) * 2 + 1
    bounds = np.zeros((out_size, 2), np.int32)
    kk = np.zeros((out_size, ksize), np.int32)
    ss = 1.0 / filterscale
    for xx in range(out_size):
        center = (xx + 0.5) * scale
        xmin = int(center - support + 0.5)
        if xmin < 0:
            xmin = 0
        xmax = int(center + support + 0.5)
        if xmax > in_size:
            xmax = in_size
        xmax -= xmin
        ww = 0.0
        k = []
        for x in range(xmax):
            a = (x + xmin - center + 0.5) * ss
            if a < 0.0:
                a = -a
            w = 1.0 - a if a < 1.0 else 0.0
            k.append(w)
            ww += w
        for x in range(xmax):
            if ww != 0.0:
                k[x] /= ww
            v = k[x] * (1 << PRECISION_BITS)
            kk[xx, x] = int(-0.5 + v) if k[x] < 0 else int(0.5 + v)
        bounds[xx] = (xmin, xmax)
    return bounds, kk, ksize


class _Tables:
    """Device-resident coefficient tables, one entry per (in_size, out_size) pair seen (a handful per run)."""

    def __init__(self):
        self.cache = {}

    def get(self, in_size, out_size, device):
        key = (in_size, out_size, device.index)
        t = self.cache.get(key)
        if t is None:
            bounds, kk, ksize = pil_bilinear_coeffs(in_size, out_size)
            t = (torch.from_numpy(bounds).to(device), torch.from_numpy(kk).to(device), ksize)
            self.cache[key] = t
        return t


_tables = _Tables()


def preprocess_images(images: torch.Tensor, out_size, flips=None, mean=_MEAN, std=_STD, jitter=None) -> torch.Tensor:
    """images: (B, Hin, Win, 3) uint8 on the GPU; out_size = (width, height); flips: (B,) uint8 (bit 0 horizontal, bit 1 vertical)
    or None; jitter: None or (order words (B,) int32, factors (B, 3) fp32) as `jitter_words` makes them.
    Returns (B, 3, height, width) fp32 = Normalize(to_tensor(flip(jitter(resize(image))))) of transforms.py:217-226."""
    L.require_cuda(images)
    if images.dtype != torch.uint8 or images.dim() != 4 or images.shape[-1] != 3:
        raise L.SdError(f"preprocess_images expects (B, H, W, 3) uint8, got {tuple(images.shape)} {images.dtype}")
    images = images.contiguous()
    B, Hin, Win, _ = images.shape
    Wout, Hout = int(out_size[0]), int(out_size[1])
    hb, hk, hks = _tables.get(Win, Wout, images.device)
    vb, vk, vks = _tables.get(Hin, Hout, images.device)
    out = torch.empty((B, 3, Hout, Wout), dtype=torch.float32, device=images.device)
    lib = L.lib()
    fl = None
    if flips is not None:
        fl = torch.as_tensor(flips, dtype=torch.uint8).to(images.device, non_blocking=True)
        if fl.numel() != B:
            raise L.SdError("flips must have one entry per image")
    m3, s3 = (C.c_float * 3)(*mean), (C.c_float * 3)(*std)
    if jitter is not None:
        order = torch.as_tensor(jitter[0], dtype=torch.int32).to(images.device, non_blocking=True)
        factors = torch.as_tensor(jitter[1], dtype=torch.float32).reshape(-1, 3).contiguous().to(images.device, non_blocking=True)
        if order.numel() != B or factors.shape[0] != B:
            raise L.SdError("jitter parameters must have one row per image")
        ws = L.workspace(lib.sd_preprocess_jitter_workspace_bytes(B, Hin, Win, Hout, Wout), images.device)
        L.check(lib.sd_preprocess_images_jitter(images.data_ptr(), B, Hin, Win, Hout, Wout, hb.data_ptr(), hk.data_ptr(), hks, vb.data_ptr(),
                                                vk.data_ptr(), vks, fl.data_ptr() if fl is not None else 0, order.data_ptr(), factors.data_ptr(),
                                                m3, s3, out.data_ptr(), ws.data_ptr(), ws.numel(), L.stream()), "sd_preprocess_images_jitter")
        return out
    ws = L.workspace(lib.sd_preprocess_workspace_bytes(B, Hin, Win, Wout), images.device)
    L.check(lib.sd_preprocess_images(images.data_ptr(), B, Hin, Win, Hout, Wout, hb.data_ptr(), hk.data_ptr(), hks, vb.data_ptr(),
                                     vk.data_ptr(), vks, fl.data_ptr() if fl is not None else 0, m3, s3, out.data_ptr(), ws.data_ptr(),
                                     ws.numel(), L.stream()), "sd_preprocess_images")
    return out


def jitter_words(order, brightness, contrast, saturation, hue):
    """One image's ColorJitter parameters in the form `sd_preprocess_images_jitter` takes: the op order (a permutation of
    0 brightness, 1 contrast, 2 saturation, 3 hue) packed two bits each, the hue shift byte `uint8(hue * 255)` torchvision's adjust_hue
    adds to the H channel (truncation toward zero, wrap-around) in bits 8-15, and the three blend factors."""
    word = sum(int(op) << (2 * k) for k, op in enumerate(order)) | ((int(hue * 255) & 0xFF) << 8)
    return word, (float(brightness), float(contrast), float(saturation))


class ValidationAugmentation:
    """transforms.py:255-261 for a batch: Resize((width, height)) + Normalize (+ the clip Encode applies, transforms.py:154)."""

    def __init__(self, args):
        self.args = args
        self.size = (args.width, args.height)

    def draws_for(self, n):
        """(flips, jitter) for n samples: validation draws nothing."""
        return None, None

    def __call__(self, images, annotations):
        """images: list of (H, W, 3) uint8 arrays / tensors (any sizes) or one (B, H, W, 3) tensor; annotations: list of
        ImageAnnotation in ORIGINAL image pixels (modified in place like the reference's Resize / flips / Encode clip do on their
        copies).  Returns ((B, 3, height, width) fp32 device tensor, annotations in network-input pixels)."""
        dev = self.args.device
        W, H = self.size
        if hasattr(images, "groups"):                                # data/feeder.py GroupedBatch: grouped by size and uploaded already
            groups = images.groups
        elif isinstance(images, torch.Tensor) and images.dim() == 4:
            groups = {tuple(images.shape[1:3]): (list(range(images.shape[0])), images)}
        else:
            by_size = {}
            for i, im in enumerate(images):
                t = im if isinstance(im, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(im))
                by_size.setdefault(tuple(t.shape[:2]), []).append((i, t))
            groups = {k: ([i for i, _ in v], torch.stack([t for _, t in v])) for k, v in by_size.items()}
        n = sum(len(idx) for idx, _ in groups.values())
        flips, jitter = self.draws_for(n)
        out = torch.empty((n, 3, H, W), dtype=torch.float32, device=dev)
        for (hin, win), (idx, stack) in groups.items():
            f = None if flips is None else [flips[i] for i in idx]
            j = None if jitter is None else ([jitter[0][i] for i in idx], [jitter[1][i] for i in idx])
            res = preprocess_images(stack.to(dev, non_blocking=True), (W, H), f, jitter=j)
            if len(groups) == 1:
                out = res
            else:
                out[torch.as_tensor(idx, device=dev)] = res
            for i in idx:
                ann = annotations[i]
                ann.img_size = ann.img_size or (win, hin)
                ann.resize((win, hin), (W, H))                       # transforms.py:58
                if flips is not None and flips[i] & 1:
                    hflip_annotation(ann, (W, H))                    # transforms.py:15 (on the resized image)
                if flips is not None and flips[i] & 2:
                    vflip_annotation(ann, (W, H))                    # transforms.py:28
                clip_annotation(ann, (W, H))                         # transforms.py:154
        return out, annotations


class TrainAugmentation(ValidationAugmentation):
    """transforms.py:211-247: Resize + RandomColorJitter + RandomHorizontalFlip + RandomVerticalFlip + Normalize, and the per-epoch
    multi-scale `trigger_random_resize` (ratios 0.75 .. 1.25 in steps of 1/16, sizes rounded down to multiples of 32).  Random decisions
    have the reference's distributions -- ColorJitter.get_params (a random order of the four ops, brightness / contrast / saturation
    factors uniform in [1 - x, 1 + x], hue in [-x, x]), `randn < prob` for each flip (transforms.py:14,27: a normal, not a uniform,
    draw: the flip probability is Phi(0.5) = 0.69), `torch.randint` for the multi-scale ratio -- from torch's global generator (the
    reference draws them inside DataLoader worker processes with per-worker seeds: its stream is not reproducible across worker counts,
    so the distributions, not a draw order, are what there is to match)."""

    ratios = (0.75, 0.8125, 0.875, 0.9375, 1, 1.0625, 1.125, 1.1875, 1.25)
    brightness, contrast, saturation, hue = 0.25, 0.25, 0.15, 0.05            # transforms.py:38

    def __init__(self, args, prob=0.5):
        super().__init__(args)
        self.prob = prob

    def draws_for(self, n):
        """(flips, jitter) of n samples from torch's global generator, in THREE vectorised draws per batch: per-sample tiny tensor ops
        (the literal form of ColorJitter.get_params + the two flip draws: seven ops per sample) cost 450 ms per batch of 64 next to
        16 busy decode threads (measured: tools/feed_probe.py), 7 ms alone.  Same distributions: a uniformly random permutation of
        the four ops (argsort of four uniforms = torch.randperm(4)), brightness / contrast / saturation uniform in [1 - x, 1 + x], hue
        uniform in [-x, x] (uniform_(a, b) = a + (b - a) * U), each flip when a standard normal draw is below `prob`."""
        if self.args.no_augmentation:
            return None, None
        u = torch.rand(n, 8, dtype=torch.float64)
        z = torch.randn(n, 2)
        orders = torch.argsort(u[:, :4], dim=1).tolist()
        lo = torch.tensor([max(0.0, 1 - self.brightness), max(0.0, 1 - self.contrast), max(0.0, 1 - self.saturation), -self.hue], dtype=torch.float64)
        hi = torch.tensor([1 + self.brightness, 1 + self.contrast, 1 + self.saturation, self.hue], dtype=torch.float64)
        vals = (lo + (hi - lo) * u[:, 4:]).tolist()
        flip_bits = ((z[:, 0] < self.prob).to(torch.int64) | ((z[:, 1] < self.prob).to(torch.int64) << 1)).tolist()
        words, factors = [], []
        for order, (b, c, s_, h) in zip(orders, vals):
            w, f3 = jitter_words(order, b, c, s_, h)
            words.append(w); factors.append(f3)
        return flip_bits, (words, factors)

    def trigger_random_resize(self):
        if self.args.no_augmentation:
            return self.size
        ratio = self.ratios[torch.randint(len(self.ratios), (1,)).item()]
        self.size = (int(ratio * self.args.width / 32) * 32, int(ratio * self.args.height / 32) * 32)
        return self.size
