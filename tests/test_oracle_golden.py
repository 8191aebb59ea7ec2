"""Pin the CPU oracle (oracle/sdnet_oracle.py) against golden vectors produced by the real
reference (tests/golden/gen_goldens.py).  CPU only."""
import numpy as np
import pytest
import torch

from oracle import sdnet_oracle as O
from tests.helpers import ENC_KEYS, objects_to_arrays, scene_from_flat

SCENES = ["scene_cfg512", "scene_small256"]


def test_prims(golden_dir):
    g = np.load(golden_dir / "prims.npz")
    sig = O.clamped_sigmoid(g["logits"])
    np.testing.assert_array_equal(sig, g["sig"])
    np.testing.assert_array_equal(O.nms(g["sig"]), g["nms"])
    # plateau: both saturated pixels survive; borders compete only with in-image neighbours
    assert g["nms"][0, 0, 5, 5] > 0 and g["nms"][0, 0, 5, 7] > 0
    assert g["nms"][1, 2, 0, 0] > 0 and g["nms"][1, 2, 39, 55] > 0
    for k in (2, 7, 40):
        s, i, c, y, x = O.topk(g["dense"], k)
        np.testing.assert_array_equal(s, g[f"topk{k}_score"])
        np.testing.assert_array_equal(i, g[f"topk{k}_ind"])
        np.testing.assert_array_equal(c, g[f"topk{k}_cls"])
        np.testing.assert_array_equal(y, g[f"topk{k}_y"])
        np.testing.assert_array_equal(x, g[f"topk{k}_x"])
    np.testing.assert_array_equal(O.transpose_and_gather(g["feat"], g["gind"]), g["gathered"])
    np.testing.assert_array_equal(O.hypot(g["hyp_in"]), g["hyp_out"])
    np.testing.assert_array_equal(O.gaussian_2d(32, 48, 17, 5, 0.1 * 32 / 3), g["gauss"])
    assert g["gauss"][5, 17] == 1.0


@pytest.mark.parametrize("tag", SCENES)
def test_encode(golden_dir, tag):
    g = np.load(golden_dir / f"{tag}.npz")
    W, H, M, N, K, P = (int(v) for v in g["cfg"])
    n = 0
    while f"scene{n}_objs" in g:
        objs = scene_from_flat(g[f"scene{n}_objs"], g[f"scene{n}_parts"])
        e = O.encode(W, H, objs, M, N, K, P, 4.0, 0.1)
        for k in ENC_KEYS:
            np.testing.assert_array_equal(e[k], g[f"enc{n}_{k}"], err_msg=f"{tag} img{n} {k}")
            assert e[k].dtype == g[f"enc{n}_{k}"].dtype
        n += 1
    assert n >= 3


def test_encode_truncation(golden_dir):
    g = np.load(golden_dir / "encode_trunc.npz")
    W, H, M, N, K, P = (int(v) for v in g["cfg"])
    for name in g["cases"]:
        objs = scene_from_flat(g[f"{name}_objs"], g[f"{name}_parts"])
        e = O.encode(W, H, objs, M, N, K, P, 4.0, 0.1)
        for k in ENC_KEYS:
            np.testing.assert_array_equal(e[k], g[f"{name}_{k}"], err_msg=f"{name} {k}")


@pytest.mark.parametrize("tag", SCENES + ["decode_thresholds"])
def test_decode(golden_dir, tag):
    """decode_thresholds: conf 0.4 / dist 0.1*64 are not fp32-representable and scores / distances are planted exactly ON
    the rounded thresholds: fp32 `>` / `<` on the device side, double compares in the host assembly."""
    g = np.load(golden_dir / f"{tag}.npz")
    W, H, M, N, K, P = (int(v) for v in g["cfg"])
    head = g["head"]
    conf, dist = (float(g["conf"]), float(g["dist"])) if "conf" in g else (0.5, 0.1)
    t = O.decode_tensors(head[:, :M], head[:, M:M + N], head[:, M + N:M + N + 2], head[:, M + N + 2:], K, P, conf, dist)
    for grp, key_out, key_inds, key_sm, n in (("anchor", "anchor_out", "anchor_inds", "anchor_scores_masked", K),
                                              ("part", "part_out", "part_inds", "part_scores_masked", P)):
        gs = g[f"dec_{grp}_score"]          # masked scores (-1 where score <= conf)
        np.testing.assert_array_equal(t[key_sm], gs)
        pos = t[key_out][..., 2] > 0        # indices are defined by the reference only where score > 0
        np.testing.assert_array_equal(t[key_inds][pos], g[f"dec_{grp}_ind"][pos])
        np.testing.assert_array_equal(t[key_out][..., 3][pos], g[f"dec_{grp}_cls"][pos])
        np.testing.assert_array_equal(t[key_out][..., 0][pos], g[f"dec_{grp}_x"][pos])
        np.testing.assert_array_equal(t[key_out][..., 1][pos], g[f"dec_{grp}_y"][pos])
        assert pos.sum() > 0
    pos = t["part_out"][..., 2] > 0
    np.testing.assert_array_equal(t["part_embeddings"][pos], g["dec_embeddings"][pos])
    out_w, out_h = W // 4, H // 4
    for b in range(head.shape[0]):
        o, p = objects_to_arrays(O.assemble_objects(t, b, conf, 4.0, out_w, out_h))
        np.testing.assert_array_equal(o, g[f"ann{b}_objs"])
        np.testing.assert_array_equal(p, g[f"ann{b}_parts"])
        r = np.array(O.raw_parts(t, b, conf, 4.0, out_w, out_h), np.float64).reshape(-1, 4)
        np.testing.assert_array_equal(r, g[f"raw{b}"])
    if tag == "decode_thresholds":
        o = g["ann0_objs"]
        assert len(o) == 3 and (o[:, 3] == float(np.float32(0.4))).sum() == 1          # the part-less anchor ON the threshold
        assert (g["raw0"][:, 3] == float(np.float32(0.4))).sum() == 1                 # the part ON the threshold stays in raw_parts


@pytest.mark.parametrize("tag", SCENES)
@pytest.mark.parametrize("fn", ["mse", "focal"])
def test_loss(golden_dir, tag, fn):
    g = np.load(golden_dir / f"{tag}.npz")
    W, H, M, N, K, P = (int(v) for v in g["cfg"])
    n_img = g["head"].shape[0]
    target = {k: np.stack([g[f"enc{n}_{k}"] for n in range(n_img)]) for k in ENC_KEYS}
    r = O.loss(g["head"], target, M, N, hm_loss_fn=fn, want_grad=True)
    ref = g[f"loss_{fn}"]
    np.testing.assert_allclose([r["total"], r["hm"], r["offset"], r["embedding"]], ref, rtol=1e-6, atol=1e-7)
    np.testing.assert_allclose(r["grad"], g[f"lossgrad_{fn}"], rtol=1e-5, atol=1e-9)


def test_fpn_head(golden_dir):
    g = np.load(golden_dir / "fpn_head.npz")
    fpn = O._Fpn(16, 8).train(); head = O._Head(8, 7)
    fpn.load_state_dict({k[4:]: torch.from_numpy(g[k]) for k in g.files if k.startswith("fpn.")})
    head.load_state_dict({k[5:]: torch.from_numpy(g[k]) for k in g.files if k.startswith("head.")})
    y = fpn(torch.from_numpy(g["x"]), torch.from_numpy(g["shortcut"]))
    np.testing.assert_allclose(y.detach().numpy(), g["fpn_out"], rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(head(y).detach().numpy(), g["head_out"], rtol=1e-5, atol=1e-6)


def test_reference_network_schema():
    """state_dict key schema of SURVEY.md A.3 (the 'adpater' spelling is load-bearing)."""
    net = O.build_reference_network(2, 1)
    sd = net.state_dict()
    assert sd["adpater.0.weight"].shape == (64, 3, 7, 7)
    assert sd["down2.0.downsample.0.weight"].shape == (128, 64, 1, 1)
    assert sd["down4.2.conv2.weight"].shape == (512, 512, 3, 3)
    assert sd["up1.weight"].shape == (128, 512, 1, 1) and sd["up1.bias"].shape == (128,)
    assert sd["up4.lateral.weight"].shape == (128, 64, 1, 1)
    assert sd["up3.conv.0.weight"].shape == (128, 128, 3, 3)
    assert sd["head.conv.weight"].shape == (7, 128, 1, 1)
    # exact counts: torchvision's published resnet34 has 21 797 672 parameters, its fc layer 512*1000 + 1000 = 513 000;
    # the trunk the reference keeps (network.py:43-50) is therefore 21 284 672, and FPN (up1 65 664 + up2 180 608 +
    # up3 164 224 + up4 156 032) + head (7*128 + 7 = 903) bring the 2-label / 1-part network to 21 852 103
    trunk = sum(p.numel() for name, p in net.named_parameters() if name.split(".")[0] in ("adpater", "down1", "down2", "down3", "down4"))
    assert trunk == 21_797_672 - 513_000 == 21_284_672
    n_params = sum(p.numel() for p in net.parameters())
    assert n_params == 21_852_103
    per_stage = {k: sum(p.numel() for name, p in net.named_parameters() if name.startswith(k + ".")) for k in ("adpater", "down1", "down2", "down3", "down4")}
    assert per_stage == {"adpater": 9_536, "down1": 221_952, "down2": 1_116_416, "down3": 6_822_400, "down4": 13_114_368}
    with torch.no_grad():
        y = net.eval()(torch.zeros(1, 3, 64, 64))
    assert y.shape == (1, 7, 16, 16)


def _resnet34_fpn_schema(M, N, depth=128):
    """Ordered (key, shape) list of the reference `Network`'s state_dict, written out from the PUBLISHED description of
    torchvision's resnet34 (BasicBlock [3, 4, 6, 3], widths 64/128/256/512; state_dict order = registration order: conv1, bn1,
    conv2, bn2, downsample) and from src/sdnet/model/network.py:41-57 -- a third statement, independent of oracle/ and of the product."""
    out = []

    def bn(prefix, c):
        out.extend([(f"{prefix}.weight", (c,)), (f"{prefix}.bias", (c,)), (f"{prefix}.running_mean", (c,)), (f"{prefix}.running_var", (c,)),
                    (f"{prefix}.num_batches_tracked", ())])

    out.append(("adpater.0.weight", (64, 3, 7, 7))); bn("adpater.1", 64)                     # network.py:43-45 (sic)
    cin = 64
    for li, (n, c) in enumerate(((3, 64), (4, 128), (6, 256), (3, 512)), start=1):          # network.py:47-50
        for b in range(n):
            p = f"down{li}.{b}"
            out.append((f"{p}.conv1.weight", (c, cin, 3, 3))); bn(f"{p}.bn1", c)
            out.append((f"{p}.conv2.weight", (c, c, 3, 3))); bn(f"{p}.bn2", c)
            if b == 0 and li > 1:
                out.append((f"{p}.downsample.0.weight", (c, cin, 1, 1))); bn(f"{p}.downsample.1", c)
            cin = c
    out += [("up1.weight", (depth, 512, 1, 1)), ("up1.bias", (depth,))]                      # network.py:52
    for name, c in (("up2", 256), ("up3", 128), ("up4", 64)):                                # network.py:53-55, 6-19
        out += [(f"{name}.lateral.weight", (depth, c, 1, 1)), (f"{name}.lateral.bias", (depth,)), (f"{name}.conv.0.weight", (depth, depth, 3, 3))]
        bn(f"{name}.conv.1", depth)
    out += [("head.conv.weight", (M + N + 4, depth, 1, 1)), ("head.conv.bias", (M + N + 4,))]  # network.py:57, 22-29
    return out


def test_torchvision_boundary_structural_pin():
    """a4 (torchvision resnet34 pieces, call sites src/sdnet/model/network.py:3,41,43-50): torchvision is absent, so arithmetic parity
    stays unpinned there; everything that CAN be pinned without it is pinned here, for the oracle AND the product's parameter tree:
    the full ordered state_dict schema (hash), BatchNorm eps / momentum, MaxPool2d(3, 2, 1), stride on conv1 of the first block,
    no conv bias in the trunk, downsample exactly where stride != 1 or cin != cout, ReLU placement, nearest x2 upsampling."""
    import hashlib
    from argparse import Namespace

    from structuredetector_amd.model import network as PN
    M, N = 2, 1
    want = _resnet34_fpn_schema(M, N)
    digest = hashlib.sha256("\n".join(f"{k} {tuple(s)}" for k, s in want).encode()).hexdigest()
    assert len(want) == 244 and digest == SCHEMA_SHA256, digest
    ref = O.ReferenceNetwork(M, N)
    prod = PN.Network(Namespace(labels={"a": 0, "b": 1}, parts={"p": 0}, fpn_depth=128), pretrained=False)
    for net in (ref, prod):
        got = [(k, tuple(v.shape)) for k, v in net.state_dict().items()]
        assert got == [(k, tuple(s)) for k, s in want]
        dtypes = {k: v.dtype for k, v in net.state_dict().items()}
        assert all(dt == (torch.int64 if k.endswith("num_batches_tracked") else torch.float32) for k, dt in dtypes.items())
    # --- oracle module hyper-parameters (what the restated trunk computes with) ---
    nn = torch.nn
    for m in ref.modules():
        if isinstance(m, nn.BatchNorm2d):
            assert m.eps == 1e-5 and m.momentum == 0.1 and m.affine and m.track_running_stats
        if isinstance(m, nn.ReLU):
            assert m.inplace
    conv1, _, relu, pool = ref.adpater
    assert (conv1.kernel_size, conv1.stride, conv1.padding, conv1.bias) == ((7, 7), (2, 2), (3, 3), None) and isinstance(relu, nn.ReLU)
    assert (pool.kernel_size, pool.stride, pool.padding, pool.dilation, pool.ceil_mode) == (3, 2, 1, 1, False)
    cin = 64
    for li, (n, c) in enumerate(((3, 64), (4, 128), (6, 256), (3, 512)), start=1):
        layer = getattr(ref, f"down{li}")
        assert len(layer) == n
        for b, blk in enumerate(layer):
            stride = 2 if (b == 0 and li > 1) else 1
            assert (blk.conv1.in_channels, blk.conv1.out_channels, blk.conv1.kernel_size, blk.conv1.stride, blk.conv1.padding) == (cin, c, (3, 3), (stride, stride), (1, 1))
            assert (blk.conv2.in_channels, blk.conv2.out_channels, blk.conv2.kernel_size, blk.conv2.stride, blk.conv2.padding) == (c, c, (3, 3), (1, 1), (1, 1))
            assert blk.conv1.bias is None and blk.conv2.bias is None
            assert (blk.downsample is not None) == (stride != 1 or cin != c)
            if blk.downsample is not None:
                ds = blk.downsample[0]
                assert (ds.kernel_size, ds.stride, ds.padding, ds.bias) == ((1, 1), (2, 2), (0, 0), None) and isinstance(blk.downsample[1], nn.BatchNorm2d)
            cin = c
    for fpn in (ref.up2, ref.up3, ref.up4):
        assert fpn.up.scale_factor == 2 and fpn.up.mode == "nearest"                        # network.py:10
        assert fpn.lateral.bias is not None and fpn.conv[0].bias is None and fpn.conv[0].padding == (1, 1)
    # BasicBlock dataflow: relu(bn2(conv2(relu(bn1(conv1(x))))) + identity_or_downsample(x)), checked on a block with hand-set weights
    blk = O._BasicBlock(4, 4, 1).eval()
    with torch.no_grad():
        for conv in (blk.conv1, blk.conv2):
            conv.weight.zero_(); conv.weight[:, :, 1, 1] = torch.eye(4)                      # identity convs
        blk.bn1.weight.fill_(-1.0)                                                           # bn1 negates: relu(-x)
    x = torch.tensor([[-2.0, -1.0, 1.0, 2.0]]).view(1, 4, 1, 1).expand(1, 4, 3, 3).contiguous()
    s = 1.0 / (1.0 + 1e-5) ** 0.5                                                            # eval BatchNorm with running (0, 1)
    want_y = torch.relu(torch.relu(-x * s) * s + x)
    np.testing.assert_allclose(blk(x).detach().numpy(), want_y.numpy(), rtol=1e-6)
    # --- product: the same hyper-parameters as constants / descriptors of the kernel schedule ---
    assert PN.BN_EPS == 1e-5 and PN.BN_MOMENTUM == 0.1
    st = prod.adpater[0]
    assert (st.k, st.stride, st.pad, st.bias) == (7, 2, 3, None)
    cin = 64
    for li, (n, c) in enumerate(((3, 64), (4, 128), (6, 256), (3, 512)), start=1):
        layer = getattr(prod, f"down{li}")
        assert len(layer) == n
        for b, blk in enumerate(layer):
            stride = 2 if (b == 0 and li > 1) else 1
            assert (blk.conv1.cin, blk.conv1.cout, blk.conv1.k, blk.conv1.stride, blk.conv1.pad, blk.conv1.bias) == (cin, c, 3, stride, 1, None)
            assert (blk.conv2.cin, blk.conv2.cout, blk.conv2.k, blk.conv2.stride, blk.conv2.pad, blk.conv2.bias) == (c, c, 3, 1, 1, None)
            assert (blk.downsample is not None) == (stride != 1 or cin != c)
            if blk.downsample is not None:
                ds = blk.downsample[0]
                assert (ds.k, ds.stride, ds.pad, ds.bias) == (1, 2, 0, None)
            cin = c


def test_resnet34_trunk_against_independent_published_port(golden_dir):
    """a4 second source (src/sdnet/model/network.py:3,41,43-50): torchvision is absent, but Hugging Face `transformers`' `ResNetModel`
    (`layer_type="basic"`, depths [3,4,6,3], 21 284 672 parameters = torchvision's resnet34 without fc) is an independent published port
    of the same network.  `tests/golden/gen_goldens.py::gen_resnet34_second_source` loaded the seeded oracle weights INTO that model and
    recorded what IT computes: the stem + max-pool output and the four stage outputs in eval and in training mode, and every running
    statistic after one training-mode forward.  The oracle's restated trunk must reproduce them: unpinned at torchvision itself,
    cross-checked against an independent port."""
    import hashlib
    g = np.load(golden_dir / "resnet34_second_source.npz")
    seed = int(g["seed"])
    net = O.build_reference_network(2, 1, seed=seed)
    sd = net.state_dict()
    gen = torch.Generator().manual_seed(seed + 1)
    for k in sorted(sd):                                                 # the generator's non-trivial running statistics
        if k.endswith("running_mean"):
            sd[k].copy_(0.2 * torch.randn(sd[k].shape, generator=gen))
        elif k.endswith("running_var"):
            sd[k].copy_(0.5 + torch.rand(sd[k].shape, generator=gen))
    h = hashlib.sha256()
    for k in sorted(k for k in sd if k.startswith(("adpater.", "down"))):
        h.update(k.encode()); h.update(sd[k].numpy().tobytes())
    assert h.hexdigest() == str(g["trunk_sha256"]), "seeded weights differ from the ones the fixture was generated with"
    x = torch.from_numpy(g["x"])

    def stages(n):
        out = [n.adpater(x)]
        for layer in (n.down1, n.down2, n.down3, n.down4):
            out.append(layer(out[-1]))
        return out

    for mode in ("eval", "train"):                                       # eval first: the training-mode pass updates the statistics
        net.train(mode == "train")
        with torch.no_grad():
            got = stages(net)
        for i, t in enumerate(got):
            want = g[f"{mode}_stage{i}"]
            assert t.shape == want.shape
            scale = np.abs(want).max()
            np.testing.assert_allclose(t.numpy(), want, rtol=0, atol=1e-5 * scale, err_msg=f"{mode} stage {i}")
    n_stats = 0
    for k, v in net.state_dict().items():
        if k.startswith(("adpater.", "down")) and k.endswith(("running_mean", "running_var", "num_batches_tracked")):
            want = g["after." + k]
            np.testing.assert_allclose(v.numpy(), want, rtol=1e-5, atol=1e-6, err_msg=k)
            n_stats += 1
    assert n_stats == 36 * 3                                             # 1 stem + 32 block + 3 downsample BatchNorms


SCHEMA_SHA256 = "df09e4ae27bca73d31fff2e2feb03c3a78219894d27e89d6d351cedad0b717cc"   # 244 entries, M=2, N=1, fpn_depth=128


def test_photometric_restatement_equals_pillow_exhaustively():
    """oracle/pil_photometric.py (what the GPU ColorJitter is checked against in its arithmetic) == the installed Pillow, bit for bit:
    all 2^24 RGB triples through convert("HSV") and convert("L"), all 2^24 HSV triples through convert("RGB"), Image.blend for every
    (degenerate, value) byte pair over factors inside / outside / at the ends of [0, 1], and the four ops on a noise image the way
    torchvision's _functional_pil.py (0.20.1) calls them (transforms.py:37-47)."""
    from PIL import Image, ImageEnhance

    from oracle import pil_photometric as PP
    v = np.arange(1 << 24, dtype=np.uint32)
    cube = np.stack([(v >> 16) & 255, (v >> 8) & 255, v & 255], -1).astype(np.uint8).reshape(4096, 4096, 3)
    img = Image.fromarray(cube, "RGB")
    assert np.array_equal(np.asarray(img.convert("HSV")), PP.rgb_to_hsv(cube))
    assert np.array_equal(np.asarray(img.convert("L")), PP.rgb_to_l(cube))
    assert np.array_equal(np.asarray(Image.fromarray(cube, "HSV").convert("RGB")), PP.hsv_to_rgb(cube))
    d, x = np.meshgrid(np.arange(256, dtype=np.uint8), np.arange(256, dtype=np.uint8), indexing="ij")
    for f in (0.0, 0.3333333, 0.75, 0.8, 0.85, 0.9123456, 1.0, 1.0000001, 1.05, 1.15, 1.249999, 1.25, 1.7):
        assert np.array_equal(PP.blend(d, x, f), np.asarray(Image.blend(Image.fromarray(d, "L"), Image.fromarray(x, "L"), f))), f
    rng = np.random.default_rng(0)
    im = rng.integers(0, 256, (97, 131, 3), dtype=np.uint8)
    pil = Image.fromarray(im)
    for f in (0.75, 0.93, 1.0, 1.17, 1.25):
        assert np.array_equal(np.asarray(ImageEnhance.Brightness(pil).enhance(f)), PP.adjust_brightness(im, f))
        assert np.array_equal(np.asarray(ImageEnhance.Contrast(pil).enhance(f)), PP.adjust_contrast(im, f))
        assert np.array_equal(np.asarray(ImageEnhance.Color(pil).enhance(f)), PP.adjust_saturation(im, f))
    for hf in (-0.05, -0.0123, 0.0, 0.02, 0.05):
        h, s_, v_ = pil.convert("HSV").split()
        nh = np.array(h, dtype=np.uint8)
        with np.errstate(over="ignore"):
            nh += np.array(hf * 255).astype(np.uint8)
        want = np.asarray(Image.merge("HSV", (Image.fromarray(nh, "L"), s_, v_)).convert("RGB"))
        assert np.array_equal(want, PP.adjust_hue(im, hf)), hf
