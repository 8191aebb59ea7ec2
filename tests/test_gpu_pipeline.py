"""GPU input pipeline: batched Resize + ColorJitter + flips + Normalize against the reference's own building blocks that
exist in this image -- PIL's Image.resize(BILINEAR) (what torchvision's F.resize calls for PIL inputs), PIL's flips, and the
to_tensor / Normalize arithmetic in torch fp32 -- bit for bit; annotation side against the product's (golden-pinned) host
functions; multi-scale training over a directory of PNG + JSON samples."""
import json

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda"
MEAN = torch.tensor([0.485, 0.456, 0.406])[:, None, None]
STD = torch.tensor([0.229, 0.224, 0.225])[:, None, None]


def reference_chain(img_u8, size, hflip=False, vflip=False):
    """transforms.py:217-226 with torchvision's functional ops written out: F.resize(PIL) -> F.hflip / F.vflip (PIL transpose) ->
    F.to_tensor (u8 / 255 in fp32) -> Normalize ((x - mean) / std in fp32)."""
    from PIL import Image
    im = Image.fromarray(img_u8).resize(size, Image.BILINEAR)
    if hflip:
        im = im.transpose(Image.FLIP_LEFT_RIGHT)
    if vflip:
        im = im.transpose(Image.FLIP_TOP_BOTTOM)
    t = torch.from_numpy(np.asarray(im).copy()).permute(2, 0, 1).to(torch.float32).div(255)
    return t.sub(MEAN).div(STD)


@pytest.mark.parametrize("shape,size", [((4, 96, 128), (64, 64)), ((3, 75, 50), (96, 64)), ((2, 64, 64), (64, 64)), ((1, 600, 800), (512, 512)),
                                        ((2, 40, 60), (160, 96))])
def test_preprocess_matches_pil_and_torch_bitwise(shape, size):
    from structuredetector_amd.data import preprocess_images
    B, H, W = shape
    rng = np.random.default_rng(H * 7 + W)
    imgs = rng.integers(0, 256, (B, H, W, 3), dtype=np.uint8)
    flips = [int(rng.integers(0, 4)) for _ in range(B)]
    got = preprocess_images(torch.from_numpy(imgs).to(DEV), size, flips).cpu()
    assert got.shape == (B, 3, size[1], size[0])
    for b in range(B):
        want = reference_chain(imgs[b], size, bool(flips[b] & 1), bool(flips[b] & 2))
        assert torch.equal(got[b], want), f"image {b} flips {flips[b]}: max diff {(got[b] - want).abs().max().item()}"
    plain = preprocess_images(torch.from_numpy(imgs).to(DEV), size).cpu()
    assert torch.equal(plain[0], reference_chain(imgs[0], size))


def test_validation_pipeline_equals_the_host_reader(golden_dir, tmp_path):
    """ValidationAugmentation on raw decoded images (GPU) == CropDataset's PIL path (CPU), images and annotations, on the 16
    evaluate16 samples of four different sizes (mixed sizes in one batch: grouped by size, order preserved)."""
    from argparse import Namespace
    from structuredetector_amd.data import CropDataset, ValidationAugmentation
    from tests.helpers import EVAL16_LABELS, EVAL16_PARTS, write_evaluate16_dir
    g = np.load(golden_dir / "evaluate16.npz")
    write_evaluate16_dir(g, tmp_path / "valid")
    args = Namespace(labels=EVAL16_LABELS, parts=EVAL16_PARTS, width=512, height=512, anchor_name="stem", device=torch.device(DEV))
    host = CropDataset(args, tmp_path / "valid")
    raw = CropDataset(args, tmp_path / "valid", raw=True)
    items = [raw[i] for i in range(8)]
    assert items[1][0].dtype == torch.uint8 and tuple(items[1][0].shape) == (480, 640, 3)
    images, anns = ValidationAugmentation(args)([im for im, _ in items], [a for _, a in items])
    assert images.shape == (8, 3, 512, 512)
    for i in range(8):
        want_img, want_ann = host[i]
        assert torch.equal(images[i].cpu(), want_img), i
        assert [(o.name, o.x, o.y, [(p.kind, p.x, p.y) for p in o.parts]) for o in anns[i].objects] == \
               [(o.name, o.x, o.y, [(p.kind, p.x, p.y) for p in o.parts]) for o in want_ann.objects]
        assert tuple(anns[i].img_size) == tuple(want_ann.img_size)


def pil_color_jitter(im, order, b, c, s, h):
    """torchvision ColorJitter.forward on a PIL image, written out with the Pillow calls its _functional_pil.py makes (0.20.1)."""
    from PIL import Image, ImageEnhance
    for fn in order:
        if fn == 0:
            im = ImageEnhance.Brightness(im).enhance(b)
        elif fn == 1:
            im = ImageEnhance.Contrast(im).enhance(c)
        elif fn == 2:
            im = ImageEnhance.Color(im).enhance(s)
        else:
            hh, ss, vv = im.convert("HSV").split()
            nh = np.array(hh, dtype=np.uint8)
            with np.errstate(over="ignore"):
                nh += np.uint8(h[0]) if isinstance(h, tuple) else np.array(h * 255).astype(np.uint8)     # (shift byte,) or the hue factor
            im = Image.merge("HSV", (Image.fromarray(nh, "L"), ss, vv)).convert("RGB")
    return im


def reference_chain_jitter(img_u8, size, jit, hflip, vflip):
    """transforms.py:217-226 in full: F.resize -> ColorJitter -> flips -> to_tensor -> Normalize, on PIL images."""
    from PIL import Image
    im = pil_color_jitter(Image.fromarray(img_u8).resize(size, Image.BILINEAR), *jit)
    if hflip:
        im = im.transpose(Image.FLIP_LEFT_RIGHT)
    if vflip:
        im = im.transpose(Image.FLIP_TOP_BOTTOM)
    t = torch.from_numpy(np.asarray(im).copy()).permute(2, 0, 1).to(torch.float32).div(255)
    return t.sub(MEAN).div(STD)


def test_color_jitter_matches_pillow_bitwise():
    """sd_preprocess_images_jitter against Pillow itself (the ops torchvision's ColorJitter calls for PIL inputs): every op alone at
    the ends and inside of its range, all 24 op orders, saturated / grey / dark images (clipping, the grey branch of the HSV round trip,
    contrast means at their rounding points), with flips; the normalised tensors must be IDENTICAL."""
    import itertools
    from structuredetector_amd.data import preprocess_images
    from structuredetector_amd.data.augment import jitter_words
    rng = np.random.default_rng(11)
    H, W, size = 72, 88, (96, 64)
    base = [rng.integers(0, 256, (H, W, 3), dtype=np.uint8),                                     # noise
            np.repeat(rng.integers(0, 256, (H, W, 1), dtype=np.uint8), 3, axis=2),               # grey: the s == 0 branch
            (rng.integers(0, 256, (H, W, 3)) // 8).astype(np.uint8),                             # dark
            np.clip(rng.integers(128, 400, (H, W, 3)), 0, 255).astype(np.uint8),                 # mostly saturated
            np.stack(np.meshgrid(np.arange(W), np.arange(H)), -1).sum(-1)[..., None].astype(np.uint8) * np.array([1, 2, 3], np.uint8)]
    cases = []
    ident = (1.0, 1.0, 1.0, 0.0)
    for k, vals in enumerate(((0.75, 0.9, 1.0, 1.1, 1.25), (0.75, 0.83, 1.0, 1.2, 1.25), (0.85, 0.97, 1.0, 1.08, 1.15), (-0.05, -0.021, 0.0, 0.013, 0.05))):
        for v in vals:
            f = list(ident); f[k] = v
            cases.append(([0, 1, 2, 3], *f))
    for order in itertools.permutations(range(4)):
        cases.append((list(order), float(rng.uniform(0.75, 1.25)), float(rng.uniform(0.75, 1.25)), float(rng.uniform(0.85, 1.15)), float(rng.uniform(-0.05, 0.05))))
    B = len(cases)
    imgs = np.stack([base[i % len(base)] for i in range(B)])
    flips = [int(rng.integers(0, 4)) for _ in range(B)]
    words, factors = zip(*(jitter_words(*c) for c in cases))
    got = preprocess_images(torch.from_numpy(imgs).to(DEV), size, flips, jitter=(list(words), list(factors))).cpu()
    for i, c in enumerate(cases):
        want = reference_chain_jitter(imgs[i], size, c, bool(flips[i] & 1), bool(flips[i] & 2))
        assert torch.equal(got[i], want), f"case {i} {c}: max diff {(got[i] - want).abs().max().item():.3e}"
    # identity parameters = the plain pipeline
    w0, f0 = jitter_words([0, 1, 2, 3], 1.0, 1.0, 1.0, 0.0)
    one = preprocess_images(torch.from_numpy(imgs[:1]).to(DEV), size, [0], jitter=([w0], [f0])).cpu()
    # (the HSV round trip is not the identity on every colour: compare with Pillow, not with the plain path)
    assert torch.equal(one[0], reference_chain_jitter(imgs[0], size, ([0, 1, 2, 3], 1.0, 1.0, 1.0, 0.0), False, False))


def test_train_augmentation_jitter_flips_and_multiscale():
    from argparse import Namespace
    from structuredetector_amd.data import TrainAugmentation
    from structuredetector_amd.utils import ImageAnnotation, Keypoint, Object
    args = Namespace(width=128, height=96, no_augmentation=False, device=torch.device(DEV))
    aug = TrainAugmentation(args)
    rng = np.random.default_rng(3)
    imgs = [rng.integers(0, 256, (60, 80, 3), dtype=np.uint8) for _ in range(6)]
    anns = [ImageAnnotation(f"{i}.png", [Object("bean", Keypoint("stem", 10.0 * i + 1, 5.0 * i + 2), [Keypoint("leaf", 70.0, 50.0)])]) for i in range(6)]
    torch.manual_seed(5)
    flips, (words, factors) = aug.draws_for(6)                      # the draws the call below will make (same generator state)
    expect = []
    for i in range(6):
        order = [(words[i] >> (2 * k)) & 3 for k in range(4)]
        assert sorted(order) == [0, 1, 2, 3] and 0.75 <= factors[i][0] <= 1.25 and 0.75 <= factors[i][1] <= 1.25 and 0.85 <= factors[i][2] <= 1.15
        expect.append(((order, *factors[i], ((words[i] >> 8) & 255,)), bool(flips[i] & 1), bool(flips[i] & 2)))
    torch.manual_seed(5)
    out, out_anns = aug(imgs, anns)
    assert any(h for _, h, _ in expect) and any(not h for _, h, _ in expect)
    assert len({tuple(j[0]) for j, _, _ in expect}) > 1
    for i, (jit, h, v) in enumerate(expect):
        assert torch.equal(out[i].cpu(), reference_chain_jitter(imgs[i], (128, 96), jit, h, v)), i
        x, y = (10.0 * i + 1) * (128 / 80), (5.0 * i + 2) * (96 / 60)         # Keypoint.resize: x *= new_w / img_w (utils.py:19-26)
        x = 128 - x - 1 if h else x
        y = 96 - y - 1 if v else y
        assert out_anns[i].objects[0].x == x and out_anns[i].objects[0].y == y
    sizes = set()
    for _ in range(40):
        w, hgt = aug.trigger_random_resize()
        assert w % 32 == 0 and hgt % 32 == 0 and 0.75 * 128 <= w <= 1.25 * 128 + 1e-9
        sizes.add((w, hgt))
    assert len(sizes) >= 3                                                    # 128 * ratio rounded down to multiples of 32: 96, 128, 160
    args.no_augmentation = True
    quiet = TrainAugmentation(args)
    assert quiet.draws_for(4) == (None, None) and quiet.trigger_random_resize() == (128, 96)
    plain, _ = quiet(imgs[:2], [ImageAnnotation(f"{i}.png", []) for i in range(2)])
    assert torch.equal(plain[0].cpu(), reference_chain(imgs[0], (128, 96)))


def test_trainer_over_a_directory_with_multiscale(golden_dir, tmp_path, monkeypatch, capsys):
    """`train --train_dir` end to end: PNG + JSON samples decoded on the host, the batch resized / flipped / normalised on the GPU,
    a different input size after each epoch (transforms.py:237-244), targets rendered for that size."""
    from structuredetector_amd.cli import train
    from tests.helpers import write_evaluate16_dir
    g = np.load(golden_dir / "evaluate16.npz")
    write_evaluate16_dir(g, tmp_path / "train")
    (tmp_path / "labels.json").write_text(json.dumps({"labels": ["bean", "maize"], "parts": ["leaf"]}))
    monkeypatch.chdir(tmp_path)
    seen = []
    from structuredetector_amd.model import trainer as T
    orig = T.TrainStep.__call__

    def spy(self, images, targets):
        seen.append((tuple(images.shape), tuple(targets["anchor_hm"].shape)))
        return orig(self, images, targets)

    monkeypatch.setattr(T.TrainStep, "__call__", spy)
    torch.manual_seed(1)
    train.main(["--train_dir", str(tmp_path / "train"), "--valid_dir", str(tmp_path / "train"), "-W", "128", "-H", "128", "-s", "stem",
                "--labels", str(tmp_path / "labels.json"), "-b", "8", "-e", "8"])
    assert len(seen) == 16 and all(s[0][0] == 8 and s[0][1] == 3 for s in seen)
    shapes = {s[0][2:] for s in seen}
    assert all(hh % 32 == 0 and ww % 32 == 0 for hh, ww in shapes) and len(shapes) >= 2, shapes
    assert all(t[2] * 4 == s[2] and t[3] * 4 == s[3] for s, t in seen)        # targets follow the epoch's input size
    assert "validation (16 images)" in capsys.readouterr().out
    assert seen[0][0][2:] == (128, 128)                                       # the first epoch runs at the configured size


def test_batch_feeder_delivers_the_dataset_bytes_in_order(golden_dir, tmp_path):
    """data/feeder.py: decode threads + pinned staging + side-stream upload.  Every batch equals the direct reads of the same samples
    (mixed image sizes in one batch: grouped, positions kept), over two passes (staging buffers recycled), with a consumer that stops
    early (the producer must not hang), and a decode failure surfaces in the consumer."""
    from argparse import Namespace
    from structuredetector_amd.data import BatchFeeder, CropDataset
    from tests.helpers import EVAL16_LABELS, EVAL16_PARTS, write_evaluate16_dir
    g = np.load(golden_dir / "evaluate16.npz")
    write_evaluate16_dir(g, tmp_path / "train")
    args = Namespace(labels=EVAL16_LABELS, parts=EVAL16_PARTS, width=128, height=128, anchor_name="stem", device=torch.device(DEV))
    ds = CropDataset(args, tmp_path / "train", raw=True)
    order = [[3, 0, 9, 12, 5], [1, 2, 15, 14, 13], [4, 6, 7, 8, 10], [11, 0, 1, 2, 3]]
    for _ in range(2):
        seen = 0
        for batch, idx in zip(BatchFeeder(ds, order, DEV, workers=4, depth=2), order):
            assert len(batch) == len(idx) and sorted(p for pos, _ in batch.groups.values() for p in pos) == list(range(len(idx)))
            for (h, w), (pos, dev) in batch.groups.items():
                assert dev.is_cuda and dev.dtype == torch.uint8 and tuple(dev.shape) == (len(pos), h, w, 3)
                for k, p in enumerate(pos):
                    want, ann = ds[idx[p]]
                    assert torch.equal(dev[k].cpu(), want)
                    assert [o.name for o in batch.annotations[p].objects] == [o.name for o in ann.objects]
            seen += 1
        assert seen == len(order)
    it = iter(BatchFeeder(ds, order * 8, DEV, workers=2, depth=2))
    next(it); next(it)
    it.close()                                                        # early stop: joins the producer thread

    class Broken:
        def __getitem__(self, j):
            if j == 7:
                raise ValueError("corrupt sample 7")
            return ds[j]
    with pytest.raises(ValueError, match="corrupt sample 7"):
        for _ in BatchFeeder(Broken(), order, DEV, workers=2):
            pass
