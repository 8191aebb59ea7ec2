"""GPU input pipeline: batched Resize + flips + Normalize against the reference's own building blocks that
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


def test_train_augmentation_flips_and_multiscale():
    from argparse import Namespace
    from structuredetector_amd.data import TrainAugmentation
    from structuredetector_amd.utils import ImageAnnotation, Keypoint, Object
    args = Namespace(width=128, height=96, no_augmentation=False, device=torch.device(DEV))
    aug = TrainAugmentation(args)
    rng = np.random.default_rng(3)
    imgs = [rng.integers(0, 256, (60, 80, 3), dtype=np.uint8) for _ in range(6)]
    anns = [ImageAnnotation(f"{i}.png", [Object("bean", Keypoint("stem", 10.0 * i + 1, 5.0 * i + 2), [Keypoint("leaf", 70.0, 50.0)])]) for i in range(6)]
    torch.manual_seed(5)
    expect = []
    for _ in range(6):                                                        # the draws TrainAugmentation makes, in its order
        h = torch.randn(1).item() < 0.5
        v = torch.randn(1).item() < 0.5
        expect.append((h, v))
    torch.manual_seed(5)
    out, out_anns = aug(imgs, anns)
    assert any(h for h, _ in expect) and any(not h for h, _ in expect)
    for i, (h, v) in enumerate(expect):
        assert torch.equal(out[i].cpu(), reference_chain(imgs[i], (128, 96), h, v)), i
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
    assert quiet.flips_for(4) is None and quiet.trigger_random_resize() == (128, 96)


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
