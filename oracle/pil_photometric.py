"""CPU restatement of the photometric ops behind the reference's RandomColorJitter -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

The reference (src/sdnet/data/transforms.py:37-47, in the default training chain :217-226) applies
`torchvision.transforms.ColorJitter(brightness=0.25, contrast=0.25, saturation=0.15, hue=0.05)` to the RESIZED PIL image.
torchvision is absent from this environment; for PIL inputs its functional ops are thin calls into Pillow
(torchvision/transforms/_functional_pil.py, 0.20.1, from the published source):

    adjust_brightness(img, f) = ImageEnhance.Brightness(img).enhance(f)
    adjust_contrast(img, f)   = ImageEnhance.Contrast(img).enhance(f)
    adjust_saturation(img, f) = ImageEnhance.Color(img).enhance(f)
    adjust_hue(img, f)        : h, s, v = img.convert("HSV").split(); h += uint8(f * 255) (wrapping); merge -> convert("RGB")

and ColorJitter.forward draws `fn_idx = torch.randperm(4)`, then brightness / contrast / saturation factors uniformly from
[max(0, 1 - x), 1 + x] and hue from [-x, x] (in that order), and applies the four ops in the order of fn_idx.

Pillow IS present (12.2), so the pin is Pillow itself: tests/test_oracle_golden.py checks every function below bit for bit against
Pillow -- all 2^24 RGB triples through convert("HSV") / convert("L"), all 2^24 HSV triples back through convert("RGB"), and the blend
for every (degenerate, value) byte pair over a set of factors.  The arithmetic below (which operations are float, which double, where
values are truncated) is what makes those exhaustive checks pass; the HIP kernel (csrc/sd_image.hip) implements the same.
"""
from __future__ import annotations

import numpy as np

BRIGHTNESS, CONTRAST, SATURATION, HUE = 0, 1, 2, 3


def rgb_to_l(rgb):
    """Pillow convert("L"): (R*19595 + G*38470 + B*7471 + 0x8000) >> 16."""
    r, g, b = (rgb[..., i].astype(np.int64) for i in range(3))
    return ((r * 19595 + g * 38470 + b * 7471 + 0x8000) >> 16).astype(np.uint8)


def blend(degenerate, image, factor):
    """Pillow Image.blend(degenerate, image, factor) on bytes: alpha is a C float; inside [0, 1] the float result is truncated,
    outside it is clipped to [0, 255] first (libImaging/Blend.c)."""
    a = np.float32(factor)
    d = degenerate.astype(np.int32)
    t = d.astype(np.float32) + a * (image.astype(np.int32) - d).astype(np.float32)          # float arithmetic throughout
    if 0.0 <= float(a) <= 1.0:
        return t.astype(np.int32).astype(np.uint8)                                            # (UINT8) cast: truncation (value is in range)
    out = np.where(t <= 0.0, 0, np.where(t >= 255.0, 255, t.astype(np.int32)))
    return out.astype(np.uint8)


def adjust_brightness(rgb, factor):
    return blend(np.zeros_like(rgb), rgb, factor)


def contrast_mean(rgb):
    """int(mean of the L image + 0.5): ImageStat sums the histogram exactly, divides in double."""
    l = rgb_to_l(rgb)
    return int(int(l.astype(np.int64).sum()) / l.size + 0.5)


def adjust_contrast(rgb, factor, mean=None):
    mean = contrast_mean(rgb) if mean is None else mean
    return blend(np.full_like(rgb, mean), rgb, factor)


def adjust_saturation(rgb, factor):
    return blend(np.repeat(rgb_to_l(rgb)[..., None], 3, axis=-1), rgb, factor)


def rgb_to_hsv(rgb):
    """Pillow convert("HSV") (libImaging/Convert.c rgb2hsv_row): float ratios, double arithmetic where the C source has double literals."""
    r, g, b = (rgb[..., i].astype(np.int32) for i in range(3))
    maxc = np.maximum(r, np.maximum(g, b)); minc = np.minimum(r, np.minimum(g, b))
    grey = maxc == minc
    cr = np.where(grey, 1, maxc - minc).astype(np.float32)
    s = cr / np.maximum(maxc, 1).astype(np.float32)
    rc = (maxc - r).astype(np.float32) / cr
    gc = (maxc - g).astype(np.float32) / cr
    bc = (maxc - b).astype(np.float32) / cr
    h = np.where(r == maxc, (bc - gc).astype(np.float32),
                 np.where(g == maxc, (2.0 + rc.astype(np.float64) - bc.astype(np.float64)).astype(np.float32),
                          (4.0 + gc.astype(np.float64) - rc.astype(np.float64)).astype(np.float32)))
    h = np.fmod(h.astype(np.float64) / 6.0 + 1.0, 1.0).astype(np.float32)
    uh = np.clip((h.astype(np.float64) * 255.0).astype(np.int32), 0, 255)
    us = np.clip((s.astype(np.float64) * 255.0).astype(np.int32), 0, 255)
    out = np.stack([np.where(grey, 0, uh), np.where(grey, 0, us), maxc], axis=-1)
    return out.astype(np.uint8)


def hsv_to_rgb(hsv):
    """Pillow HSV -> RGB (Convert.c hsv2rgb): i = floor(h * 6 / 255), f the remainder, p / q / t rounded half away from zero."""
    h, s, v = (hsv[..., i].astype(np.int32) for i in range(3))
    hf = h.astype(np.float32).astype(np.float64) * 6.0 / 255.0
    i = np.floor(hf).astype(np.int32)
    f = (hf - i.astype(np.float32).astype(np.float64)).astype(np.float32)
    fs = (s.astype(np.float32).astype(np.float64) / 255.0).astype(np.float32)
    vf = v.astype(np.float32).astype(np.float64)
    rnd = lambda x: np.floor(x + 0.5).astype(np.int32)                                        # C round() of a non-negative double
    p = np.clip(rnd(vf * (1.0 - fs.astype(np.float64))), 0, 255)
    q = np.clip(rnd(vf * (1.0 - fs.astype(np.float64) * f.astype(np.float64))), 0, 255)
    t = np.clip(rnd(vf * (1.0 - fs.astype(np.float64) * (1.0 - f.astype(np.float64)))), 0, 255)
    k = i % 6
    r = np.choose(k, [v, q, p, p, t, v]); g = np.choose(k, [t, v, v, q, p, p]); b = np.choose(k, [p, p, t, v, v, q])
    grey = s == 0
    return np.stack([np.where(grey, v, r), np.where(grey, v, g), np.where(grey, v, b)], axis=-1).astype(np.uint8)


def hue_shift_byte(factor):
    """np.uint8(factor * 255) as torchvision adds it to the H channel: the double product truncated toward zero, then wrapped modulo 256."""
    return int(factor * 255) & 0xFF


def adjust_hue(rgb, factor):
    hsv = rgb_to_hsv(rgb)
    hsv[..., 0] = (hsv[..., 0].astype(np.int32) + hue_shift_byte(factor)).astype(np.uint8)     # uint8 wrap-around
    return hsv_to_rgb(hsv)


def color_jitter(rgb, order, brightness, contrast, saturation, hue):
    """torchvision ColorJitter.forward on one (H, W, 3) uint8 image: the four ops in the order of `order` (a permutation of 0..3)."""
    for fn in order:
        if fn == BRIGHTNESS:
            rgb = adjust_brightness(rgb, brightness)
        elif fn == CONTRAST:
            rgb = adjust_contrast(rgb, contrast)
        elif fn == SATURATION:
            rgb = adjust_saturation(rgb, saturation)
        else:
            rgb = adjust_hue(rgb, hue)
    return rgb
