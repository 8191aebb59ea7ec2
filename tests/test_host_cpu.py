"""CPU-only tests: the C-ABI library loads and exports every declared symbol, host-side logic
(value types, Encode's float64 slot planner, LossStats, flag table) behaves like the reference.
No GPU compute is issued here."""
import json
import os
import re
import subprocess
import sys
from argparse import Namespace
from pathlib import Path

import numpy as np
import pytest

from tests.helpers import scene_from_flat

ROOT = Path(__file__).resolve().parent.parent


def make_args(M, N, K, P, **kw):
    labels = {f"label{i}": i for i in range(M)}
    parts = {f"part{i}": i for i in range(N)}
    d = dict(labels=labels, parts=parts, _r_labels={v: k for k, v in labels.items()},
             _r_parts={v: k for k, v in parts.items()}, anchor_name="stem", down_ratio=4.0, max_objects=K, max_parts=P,
             conf_threshold=0.5, decoder_dist_thresh=0.1, sigma_gauss=0.1, hm_loss_fn="mse", hm_weight=1.0,
             offset_weight=0.001, embedding_weight=0.001, fpn_depth=128)
    d.update(kw)
    return Namespace(**d)


def to_annotation(args, objs, name="img.png"):
    from structuredetector_amd.utils import ImageAnnotation, Keypoint, Object
    return ImageAnnotation(name, [Object(args._r_labels[l], Keypoint(args.anchor_name, x, y),
                                         [Keypoint(args._r_parts[k], px, py) for (k, px, py) in ps])
                                  for (l, x, y, ps) in objs])


def test_library_exports_every_declared_symbol():
    from structuredetector_amd import _lib as L
    header = (ROOT / "include" / "sdnet_hip.h").read_text()
    header = re.sub(r"/\*.*?\*/", "", header, flags=re.S)
    declared = set(re.findall(r"\b(sd_[a-z0-9_]+)\s*\(", header))
    assert declared, "no declarations parsed"
    handle = L.lib()
    for name in declared:
        assert hasattr(handle, name), f"{name} declared in include/sdnet_hip.h but not exported"
    assert declared == set(L.declared_symbols()), declared ^ set(L.declared_symbols())
    assert handle.sd_version() >= 1
    assert handle.sd_build_flags() == 0          # no SD_ABLATE_* (wrong-results timing switches) in the shipped library


def test_ops_refuse_cpu_tensors():
    import torch
    from structuredetector_amd import _lib as L
    from structuredetector_amd.utils import nms
    with pytest.raises(L.SdError):
        nms(torch.zeros(1, 1, 8, 8))


def test_value_types_roundtrip(tmp_path):
    from structuredetector_amd.utils import ImageAnnotation, Keypoint, Object, Box
    ann = ImageAnnotation("a/b.png", [Object("bean", Keypoint("stem", 10.0, 20.0), [Keypoint("leaf", 12.5, 18.0, 0.9)],
                                             Box(1, 2, 30, 40))], img_size=[100, 200])
    r = ann.resized((100, 200), (50, 50))
    assert ann.objects[0].x == 10.0 and r.objects[0].x == 5.0 and r.objects[0].y == 5.0
    assert r.objects[0].parts[0].x == 6.25 and r.objects[0].box.x_max == 15.0
    n = ann.normalized()
    assert n.objects[0].anchor.x == 0.1 and n.objects[0].anchor.y == 0.1
    p = tmp_path / "x.json"
    d = ann.json_repr()
    assert d["objects"][0]["parts"][0]["kind"] == "stem" and d["objects"][0]["parts"][1]["score"] == 0.9
    p.write_text(json.dumps(d))
    back = ImageAnnotation.from_json(p, "stem")
    assert back.objects[0].name == "bean" and back.objects[0].parts[0].kind == "leaf" and back.nb_parts == 1
    assert len(back) == 1 and not back.is_empty and back.objects[0].box.width == 29
    with pytest.raises(AssertionError):
        Object.from_json({"label": "x", "box": None, "parts": []}, "stem")
    assert abs(Keypoint("a", 0, 0).distance(Keypoint("b", 3, 4)) - 5.0) < 1e-12


@pytest.mark.parametrize("tag", ["scene_cfg512", "scene_small256"])
def test_encode_plan_matches_reference(golden_dir, tag):
    """Host stage of Encode (float64 clip/resize/truncate -> inds, offsets, embeddings, masks): bit-exact."""
    from structuredetector_amd.data import Encode
    g = np.load(golden_dir / f"{tag}.npz")
    W, H, M, N, K, P = (int(v) for v in g["cfg"])
    args = make_args(M, N, K, P)
    enc = Encode(args)
    from structuredetector_amd.data.transforms import scenes_to_flat
    n_img = g["head"].shape[0]
    anns = [to_annotation(args, scene_from_flat(g[f"scene{n}_objs"], g[f"scene{n}_parts"])) for n in range(n_img)]
    plan = enc.plan(W, H, *scenes_to_flat(anns, args.labels, args.parts))
    B = n_img
    f32, i64, u8 = plan["f32"], plan["i64"], plan["u8"]
    got = {"anchor_inds": i64[:B * K].reshape(B, K), "part_inds": i64[B * K:].reshape(B, P),
           "anchor_offsets": f32[:B * K * 2].reshape(B, K, 2), "part_offsets": f32[B * K * 2:B * K * 2 + B * P * 2].reshape(B, P, 2),
           "embeddings": f32[B * K * 2 + B * P * 2:].reshape(B, P, 2),
           "anchor_mask": u8[:B * K].reshape(B, K).astype(bool), "part_mask": u8[B * K:].reshape(B, P).astype(bool)}
    for k, v in got.items():
        ref = np.stack([g[f"enc{n}_{k}"] for n in range(n_img)])
        np.testing.assert_array_equal(v, ref, err_msg=k)
    # renderer inputs: CSR over (image, channel) and the centre pixels equal the flat indices
    n = plan["n_kp"]
    cx, cy, ptr = plan["i32"][:n], plan["i32"][n:2 * n], plan["i32"][2 * n:]
    assert ptr[0] == 0 and ptr[-1] == n and (np.diff(ptr) >= 0).all()
    assert n == got["anchor_mask"].sum() + got["part_mask"].sum()
    assert plan["two_sigma2"] == float(np.float32(2 * (0.1 * min(W // 4, H // 4) / 3) ** 2))


def test_encode_plan_truncation(golden_dir):
    from structuredetector_amd.data import Encode
    from structuredetector_amd.data.transforms import scenes_to_flat
    g = np.load(golden_dir / "encode_trunc.npz")
    W, H, M, N, K, P = (int(v) for v in g["cfg"])
    args = make_args(M, N, K, P)
    for name in g["cases"]:
        ann = to_annotation(args, scene_from_flat(g[f"{name}_objs"], g[f"{name}_parts"]))
        from structuredetector_amd.utils import clip_annotation
        clip_annotation(ann, (W, H))
        plan = Encode(args).plan(W, H, *scenes_to_flat([ann], args.labels, args.parts))
        np.testing.assert_array_equal(plan["i64"][:K], g[f"{name}_anchor_inds"], err_msg=name)
        np.testing.assert_array_equal(plan["i64"][K:], g[f"{name}_part_inds"], err_msg=name)
        np.testing.assert_array_equal(plan["u8"][:K].astype(bool), g[f"{name}_anchor_mask"], err_msg=name)
        np.testing.assert_array_equal(plan["u8"][K:].astype(bool), g[f"{name}_part_mask"], err_msg=name)
        np.testing.assert_array_equal(plan["f32"][:2 * K].reshape(K, 2), g[f"{name}_anchor_offsets"], err_msg=name)
        np.testing.assert_array_equal(plan["f32"][2 * K + 2 * P:].reshape(P, 2), g[f"{name}_embeddings"], err_msg=name)


def test_loss_stats_arithmetic():
    from structuredetector_amd.model import LossStats
    s = LossStats(1.0, 2.0, 3.0)
    assert s.total_loss == 6.0
    t = s + LossStats(1.0, 1.0, 1.0)
    assert (t.hm_loss, t.offset_loss, t.embedding_loss) == (2.0, 3.0, 4.0)
    t /= 2
    assert t.offset_loss == 1.5
    s += t
    assert s.hm_loss == 2.0
    s.reset()
    assert s.total_loss == 0.0


def test_flag_table_matches_reference_surface():
    from structuredetector_amd.utils.args import Arguments
    p = Arguments().parser
    ns = p.parse_args([])
    expect = dict(labels="labels.json", anchor_name="anchor", width=512, height=512, in_channels=3, fpn_depth=128,
                  pretrained_model=None, batch_size=8, epochs=100, no_augmentation=False, learning_rate=1e-3, lr_step=3,
                  down_ratio=4.0, hm_loss_fn="mse", max_objects=20, max_parts=40, hm_weight=1.0, offset_weight=0.001,
                  embedding_weight=0.001, sigma_gauss=0.1, conf_threshold=0.5, dist_threshold=0.05,
                  decoder_dist_thresh=0.1, csi_threshold=0.75, csv_path=None, use_amp=False)
    for k, v in expect.items():
        assert getattr(ns, k) == v, k
    ns = p.parse_args("-W 256 -H 320 -b 4 -n 7 -k 9 -t 0.3 -o m.pth -s stem -f focal".split())
    assert (ns.width, ns.height, ns.batch_size, ns.max_objects, ns.max_parts) == (256, 320, 4, 7, 9)
    assert ns.conf_threshold == 0.3 and ns.pretrained_model == "m.pth" and ns.anchor_name == "stem" and ns.hm_loss_fn == "focal"


def test_c_abi_rejects_bad_arguments_without_touching_the_gpu():
    """Every entry point validates before it launches: invalid calls return a negative SD_ERR_* code and set
    sd_last_error(); the *_workspace_bytes queries are pure host arithmetic."""
    import ctypes as C
    from structuredetector_amd import _lib as L
    lib = L.lib()
    assert lib.sd_clamped_sigmoid(0, 0, 16, 0) == -1 and b"sd_clamped_sigmoid" in lib.sd_last_error()
    assert lib.sd_clamped_sigmoid(8, 24, 16, 0) == -3                      # misaligned pointers -> SD_ERR_ALIGN
    assert lib.sd_topk(0, 0, 0, 1, 1, 8, 8, 4, 0, 0, 0, 0, 0, 0, 0, 0) == -1
    assert lib.sd_decode_peaks(16, 64, 64, 1, 1, 8, 8, 4096, 16, 16, 16, 16, 16, 16, 1 << 20, 0) == -1   # k > SD_MAX_TOPK
    assert b"out of range" in lib.sd_last_error()
    assert lib.sd_render_targets(16, 16, 16, 1, 3, 8, 6, C.c_float(1.0), 16, 0) == -1                    # w % 4 != 0
    assert lib.sd_adam_step(16, 16, 16, 16, 10, 1, C.c_float(1e-3), C.c_float(0.9), C.c_float(0.999), C.c_float(1e-8), C.c_float(1.0), 0) == -1
    d = L.ConvDesc()
    d.B, d.Hi, d.Wi, d.Cin, d.Cout, d.R, d.S, d.stride, d.pad, d.Ho, d.Wo = 1, 8, 8, 48, 64, 3, 3, 1, 1, 8, 8
    assert lib.sd_conv2d_fwd(16, 16, 16, C.byref(d), 0, 0, 0, 0, 0, 0, 0, 0) == -1 and b"Cin" in lib.sd_last_error()
    d.Cin, d.Ho = 64, 7
    assert lib.sd_conv2d_fwd(16, 16, 16, C.byref(d), 0, 0, 0, 0, 0, 0, 0, 0) == -1 and b"geometry" in lib.sd_last_error()
    assert lib.sd_bn_apply(16, 16, 100, 6, 16, 16, 16, 16, 0, 1, 0, 0) == -1                                  # C % 4 != 0
    # workspace queries
    assert lib.sd_decode_workspace_bytes(64, 2, 1, 128, 128, 20, 40) >= 64 * 3 * 128 * 128 * 8
    assert lib.sd_decode_packed_words(64, 20, 40) == 64 * (6 * 20 + 11 * 40 + 1)
    d.Ho = 8
    assert lib.sd_conv2d_wgrad_workspace_bytes(C.byref(d)) >= 64 * 9 * 64 * 4
    assert lib.sd_loss_workspace_bytes(64, 2, 1, 128, 128) > 0


def test_set_option_is_host_only_and_rejects_unknown_names():
    from structuredetector_amd import _lib as L
    lib = L.lib()
    assert lib.sd_set_option(b"conv_patch_min_tiles", 512) == 0
    assert lib.sd_set_option(b"conv_patch_bn64", 0) == 0
    assert lib.sd_set_option(b"no_such_option", 1) == -1
    assert b"no_such_option" in lib.sd_last_error()
    # the decoder's tuning knob: sizes handed out cover both tile heights, whatever the knob says at launch time
    ws = lib.sd_decode_fused_workspace_bytes(64, 2, 1, 132, 128, 20, 40)
    assert lib.sd_decode_set_option(b"tall_tiles_from", 1) == 0
    assert lib.sd_decode_fused_workspace_bytes(64, 2, 1, 132, 128, 20, 40) == ws >= 64 * 3 * 2 * 5 * 2048 * 8
    assert lib.sd_decode_set_option(b"tall_tiles_from", 2688) == 0
    assert lib.sd_decode_set_option(b"nope", 1) == -1 and b"nope" in lib.sd_last_error()
    # kernel choice is host arithmetic on the descriptor: a 3x3 / 1 / 1 conv with a chip-filling grid takes the patch kernel,
    # a strided one the 256-row tile kernel, a small one the 128-row tiles
    d = L.ConvDesc()
    d.B, d.Hi, d.Wi, d.Cin, d.Cout, d.R, d.S, d.stride, d.pad, d.Ho, d.Wo = 64, 64, 64, 128, 128, 3, 3, 1, 1, 64, 64
    import ctypes as C
    assert lib.sd_conv2d_kernel_name(C.byref(d), 0) == b"k_conv3x3_patch<128, false>"
    assert lib.sd_conv2d_kernel_name(C.byref(d), 1) == b"k_conv3x3_patch<128, false>"
    assert lib.sd_conv2d_kernel_name(C.byref(d), 2) == b"k_wgrad3x3_ring2"
    assert lib.sd_set_option(b"wgrad_f32_ring", 0) == 0 and lib.sd_conv2d_kernel_name(C.byref(d), 2) == b"k_wgrad3x3<32>"
    assert lib.sd_set_option(b"wgrad_f32_ring", 1) == 0 and lib.sd_conv2d_kernel_name(C.byref(d), 2) == b"k_wgrad3x3_ring"
    assert lib.sd_set_option(b"wgrad_f32_ring", 2) == 0
    # layer1 at bs=64 (128x128 maps, 64 channels): the fp32 row stream (256 units of 64 rows); small batches keep the tile kernel
    d.Hi = d.Wi = d.Ho = d.Wo = 128; d.Cin = d.Cout = 64
    assert lib.sd_conv2d_kernel_name(C.byref(d), 0) == b"k_conv3x3_c64_rows_f32"
    assert lib.sd_conv2d_kernel_name(C.byref(d), 1) == b"k_conv3x3_c64_rows_f32"
    d.B = 8
    assert lib.sd_conv2d_kernel_name(C.byref(d), 0) == b"k_conv_igemm<64, 0, false>"
    d.B = 64
    # layer4 at bs=64 (16x16 maps, 512 channels): 256 patch tiles of 128 channels do not fill the chip, 512 tiles of 64 channels do
    d.Hi = d.Wi = d.Ho = d.Wo = 16; d.Cin = d.Cout = 512
    assert lib.sd_conv2d_kernel_name(C.byref(d), 0) == b"k_conv3x3_patch<64, false>"
    assert lib.sd_conv2d_kernel_name(C.byref(d), 1) == b"k_conv3x3_patch<64, false>"
    assert lib.sd_set_option(b"conv_patch_narrow", 0) == 0
    assert lib.sd_conv2d_kernel_name(C.byref(d), 0) == b"k_conv_igemm<128, 0, false>"
    assert lib.sd_set_option(b"conv_patch_narrow", 2) == 0
    d.Hi = d.Wi = d.Ho = d.Wo = 64; d.Cin = d.Cout = 128
    d.stride, d.Ho, d.Wo = 2, 32, 32
    assert lib.sd_conv2d_kernel_name(C.byref(d), 1) == b"k_conv_igemm_big<128, 2, false>"
    d.B, d.stride, d.Ho, d.Wo = 1, 1, 64, 64
    assert lib.sd_conv2d_kernel_name(C.byref(d), 1) == b"k_conv_igemm<128, 0, false>"
    # the bf16 dispatch (passes 16 / 17 = sd_conv2d_fwd_bf16 / sd_conv2d_dgrad_bf16) at bs=64, 512x512: layer2 / layer3 3x3 convs take the
    # two-group kernel (>= 200 tiles of 512 pixels), layer1 the row stream (>= 16 rows per unit), layer4 (128 tiles) 64-channel patch tiles, strided convs the
    # generic implicit GEMM, up4.conv (128-wide map) the two-group kernel on column strips (the one-group patch kernel without them); small batches fall back to the tile kernels / split-K
    def name(B, H, cin, cout, k=3, stride=1, pad=1, which=16):
        e = L.ConvDesc()
        e.B, e.Hi, e.Wi, e.Cin, e.Cout, e.R, e.S, e.stride, e.pad = B, H, H, cin, cout, k, k, stride, pad
        e.Ho = e.Wo = (H + 2 * pad - k) // stride + 1
        return lib.sd_conv2d_kernel_name(C.byref(e), which).decode()
    for which in (16, 17):
        assert name(64, 64, 128, 128, which=which) == "k_conv3x3_bf16_pp"
        assert name(64, 32, 256, 256, which=which) == "k_conv3x3_bf16_pp"
        assert name(64, 128, 64, 64, which=which) == "k_conv3x3_c64_rows16_bf16"
        assert name(64, 16, 512, 512, which=which) == "k_conv3x3_patch<64, true>"      # layer4: 128 two-group tiles, 256 patch tiles of 128 channels, 512 of 64
        assert name(64, 128, 128, 128, which=which) == "k_conv3x3_bf16_pp"          # up4.conv: 64-pixel column strips
        assert lib.sd_set_option(b"conv_pp_strips", 0) == 0
        assert name(64, 128, 128, 128, which=which) == "k_conv3x3_patch_roll<128, true>"      # (128-wide map: the rolling-buffer entry point)
        assert lib.sd_set_option(b"conv_pp_strips", 1) == 0
    assert name(64, 64, 128, 256, stride=2) == "k_conv_igemm<128, 0, true>"
    assert name(64, 64, 128, 256, stride=2, which=17) == "k_conv_igemm<128, 2, true>"
    assert name(16, 64, 128, 128) == "k_conv3x3_patch<64, true>"          # 128 two-group tiles < 200, 256 one-group tiles < 512 <= 512 of 64 channels
    assert name(8, 64, 128, 128) == "k_conv_igemm<128, 0, true>"          # ... and 256 of 64 channels < 512
    assert name(32, 64, 128, 128) == "k_conv3x3_bf16_pp"                  # 256 two-group tiles
    assert name(16, 128, 64, 64) == "k_conv3x3_patch_roll<64, true>"      # 8 rows per unit < 16; 128-wide map: rolling-buffer entry point
    assert name(16, 256, 64, 64) == "k_conv3x3_c64_rows16_bf16"             # stress config (1024 x 1024 inputs): two strips per row
    assert name(1, 64, 128, 128) == "k_conv_igemm<128, 0, true>"          # bs=1: split-K
    # round 4: the FPN laterals (1x1 / stride 1 onto 128 channels from 64 / 128) stream from 65536 output pixels; the 128 -> 128 1x1
    # data-gradient too; other widths and small maps keep the tile kernel; the option moves the threshold
    assert name(64, 128, 64, 128, k=1, pad=0) == "k_conv1x1_stream_bf16<64, 2>"
    assert name(64, 64, 128, 128, k=1, pad=0) == "k_conv1x1_stream_bf16<128, 1>"
    assert name(64, 64, 128, 128, k=1, pad=0, which=17) == "k_conv1x1_stream_bf16<128, 1>"
    assert name(64, 32, 256, 128, k=1, pad=0) == "k_conv_igemm<128, 0, true>"
    assert name(1, 128, 64, 128, k=1, pad=0) == "k_conv_igemm<128, 0, true>"
    assert lib.sd_set_option(b"conv1x1_stream_min_pixels", 1 << 30) == 0
    assert name(64, 128, 64, 128, k=1, pad=0) == "k_conv_igemm<128, 0, true>"
    assert lib.sd_set_option(b"conv1x1_stream_min_pixels", 32 * 2048) == 0
    assert lib.sd_set_option(b"wgrad_bf16_ring", 5) == 0
    # the head fused into the last FPN conv (inference): where that conv takes the two-group kernel, up to 32 head channels
    def head_ok(B, H, co, cout=128):
        e = L.ConvDesc()
        e.B, e.Hi, e.Wi, e.Cin, e.Cout, e.R, e.S, e.stride, e.pad, e.Ho, e.Wo = B, H, H, 128, cout, 3, 3, 1, 1, H, H
        return lib.sd_conv2d_fwd_bf16_head_supported(C.byref(e), co)
    assert head_ok(64, 128, 7) == 1 and head_ok(16, 256, 20) == 1 and head_ok(64, 128, 32) == 1
    assert head_ok(1, 128, 7) == 0 and head_ok(64, 128, 33) == 0 and head_ok(64, 128, 7, cout=256) == 0
    assert lib.sd_head_split_bf16_bytes() == 2 * 32 * 128 * 2 + 32 * 4
    assert lib.sd_head_split_bf16(16, 16, 40, 16, 0) == -1 and b"head_co" in lib.sd_last_error()
    e = L.ConvDesc()
    e.B, e.Hi, e.Wi, e.Cin, e.Cout, e.R, e.S, e.stride, e.pad, e.Ho, e.Wo = 1, 128, 128, 128, 128, 3, 3, 1, 1, 128, 128
    assert lib.sd_conv2d_fwd_bf16_head(16, 16, C.byref(e), 0, 0, 1, 16, 7, 16, 0) == -1          # bs = 1: not supported -> refused before any launch


def test_network_parameter_count_matches_published_resnet34():
    """a4 pin: the trunk the reference keeps from torchvision's resnet34 (network.py:43-50) has 21 797 672 - 513 000 (fc)
    = 21 284 672 parameters; FPN + head bring the 2-label / 1-part network to 21 852 103 (21.852 M)."""
    from structuredetector_amd.model import Network
    net = Network(Namespace(labels={"bean": 0, "maize": 1}, parts={"leaf": 0}, fpn_depth=128), pretrained=False)
    trunk = sum(p.numel() for name, p in net.named_parameters() if name.split(".")[0] in ("adpater", "down1", "down2", "down3", "down4"))
    assert trunk == 21_284_672
    assert sum(p.numel() for p in net.parameters()) == 21_852_103
    from oracle import sdnet_oracle as O
    ref = O.build_reference_network(2, 1)
    assert {k: tuple(v.shape) for k, v in net.state_dict().items()} == {k: tuple(v.shape) for k, v in ref.state_dict().items()}


def test_steplr_matches_torch_scheduler():
    """StepLR (trainer.py:54-56: torch.optim.lr_scheduler.StepLR(optimizer, step_size=args.lr_step), gamma 0.1) against torch's."""
    import torch
    from structuredetector_amd.model.trainer import StepLR

    class _Step:
        lr = 1e-3

    for step_size in (1, 3, 33):
        p = torch.nn.Parameter(torch.zeros(1))
        opt = torch.optim.Adam([p], 1e-3)
        ref = torch.optim.lr_scheduler.StepLR(opt, step_size=step_size)
        obj = _Step(); obj.lr = 1e-3
        mine = StepLR(obj, step_size)
        for epoch in range(100):
            opt.step(); ref.step(); mine.step()
            assert abs(obj.lr - opt.param_groups[0]["lr"]) <= 1e-12 * obj.lr, (step_size, epoch)
        state = mine.state_dict()
        other = StepLR(_Step(), 7)
        other.load_state_dict(state)
        assert other.epoch == 100 and abs(other.step_obj.lr - obj.lr) < 1e-30


def test_draw_marks_anchors_parts_and_links():
    """utils/visualization.py::draw (reference visualization.py:13-50): discs of 1 % of the short side in the xxh64-derived colours
    (args.py:264-267), a white link per part."""
    from PIL import Image
    from structuredetector_amd.utils import ImageAnnotation, Keypoint, Object, draw, draw_keypoints, get_unique_color_map
    args = Namespace(labels={"bean": 0, "maize": 1}, parts={"leaf": 0})
    args._label_color_map = get_unique_color_map(args.labels)
    args._part_color_map = get_unique_color_map(args.parts)
    assert args._label_color_map["bean"] == (28, 48, 154) and args._part_color_map["leaf"] == (148, 24, 183)   # first 3 bytes of xxh64(name)
    ann = ImageAnnotation("x.jpg", [Object("bean", Keypoint("stem", 50.0, 60.0, 0.9), [Keypoint("leaf", 150.0, 60.0, 0.8)]),
                                    Object("maize", Keypoint("stem", 100.0, 150.0, 0.7), [])])
    base = Image.new("RGB", (200, 300), (0, 0, 0))
    out = draw(base, ann, args)
    assert out.size == (200, 300) and base.getpixel((50, 60)) == (0, 0, 0)                 # the input image is not modified
    assert out.getpixel((50, 60)) == args._label_color_map["bean"] and out.getpixel((100, 150)) == args._label_color_map["maize"]
    assert out.getpixel((150, 60)) == args._part_color_map["leaf"] and out.getpixel((100, 60)) == (255, 255, 255)   # link
    assert out.getpixel((50, 64)) == (0, 0, 0)                                              # radius = int(200 * 1 / 100) = 2
    kp = draw_keypoints(base, [Keypoint("maize", 20, 20), Keypoint("leaf", 40, 40)], args)
    assert kp.getpixel((20, 20)) == args._label_color_map["maize"] and kp.getpixel((40, 40)) == args._part_color_map["leaf"]
    import torch
    t = torch.zeros(3, 40, 60)                                                             # normalised tensor input: un-normalised first
    img = draw(t, ImageAnnotation("t", []), args)
    assert img.size == (60, 40) and img.getpixel((5, 5)) == (124, 116, 104)                # round(255 * ImageNet mean)


def test_annotation_transforms_vs_reference(golden_dir):
    """clip / hflip / vflip / resize of annotations incl. boxes (utils.py:19-26,364-415) and the colour map (utils.py:476-479) against
    the reference's own functions (tests/golden/annotation_transforms.npz)."""
    import copy
    from structuredetector_amd.utils import (Box, ImageAnnotation, Keypoint, Object, clip_annotation, get_unique_color_map,
                                             hflip_annotation, vflip_annotation)
    g = np.load(golden_dir / "annotation_transforms.npz")

    def build(rows):
        objs, i = [], 0
        while i < len(rows):
            x, y, b0, b1, b2, b3, n = rows[i]
            n = int(n)
            parts = [Keypoint("leaf", rows[i + 1 + j][0], rows[i + 1 + j][1]) for j in range(n)]
            objs.append(Object("bean", Keypoint("stem", x, y), parts, None if np.isnan(b0) else Box(b0, b1, b2, b3)))
            i += 1 + n
        return ImageAnnotation("a.jpg", objs)

    def flat(a):
        rows = []
        for o in a.objects:
            b = o.box
            rows.append([o.x, o.y] + ([b.x_min, b.y_min, b.x_max, b.y_max] if b is not None else [np.nan] * 4) + [len(o.parts)])
            rows += [[k.x, k.y, np.nan, np.nan, np.nan, np.nan, -1] for k in o.parts]
        return np.array(rows, np.float64)

    ann = build(g["input"])
    np.testing.assert_array_equal(flat(ann), g["input"])
    np.testing.assert_array_equal(flat(hflip_annotation(copy.deepcopy(ann), (200, 100))), g["hflip"])
    np.testing.assert_array_equal(flat(vflip_annotation(copy.deepcopy(ann), (200, 100))), g["vflip"])
    np.testing.assert_array_equal(flat(vflip_annotation(hflip_annotation(copy.deepcopy(ann), (200, 100)), (200, 100))), g["hvflip"])
    np.testing.assert_array_equal(flat(clip_annotation(copy.deepcopy(ann), (200, 100))), g["clip"])
    np.testing.assert_array_equal(flat(copy.deepcopy(ann).resize((200, 100), (512, 384))), g["resized"])
    cm = get_unique_color_map([str(n) for n in g["color_names"]])
    assert [list(cm[str(n)]) for n in g["color_names"]] == g["colors"].tolist()


def test_pil_bilinear_coefficient_tables_reproduce_pillow():
    """The host half of the GPU Resize: coefficient tables of Pillow's separable 8-bit resampling (data/augment.py) applied with
    numpy must give exactly the bytes of Image.resize(BILINEAR) -- up-scaling, antialiased down-scaling, identity passes."""
    from PIL import Image
    from structuredetector_amd.data.augment import PRECISION_BITS, pil_bilinear_coeffs

    def resample(img, Wout, Hout):
        Hin, Win, _ = img.shape
        hb, hk, _ = pil_bilinear_coeffs(Win, Wout)
        vb, vk, _ = pil_bilinear_coeffs(Hin, Hout)
        clip8 = lambda v: np.clip(v >> PRECISION_BITS, 0, 255).astype(np.uint8)
        tmp = np.zeros((Hin, Wout, 3), np.uint8)
        for x in range(Wout):
            acc = np.full((Hin, 3), 1 << (PRECISION_BITS - 1), np.int64)
            for t in range(hb[x, 1]):
                acc += img[:, hb[x, 0] + t].astype(np.int64) * int(hk[x, t])
            tmp[:, x] = clip8(acc)
        out = np.zeros((Hout, Wout, 3), np.uint8)
        for y in range(Hout):
            acc = np.full((Wout, 3), 1 << (PRECISION_BITS - 1), np.int64)
            for t in range(vb[y, 1]):
                acc += tmp[vb[y, 0] + t].astype(np.int64) * int(vk[y, t])
            out[y] = clip8(acc)
        return out

    rng = np.random.default_rng(0)
    for (Hin, Win, Hout, Wout) in [(48, 64, 32, 32), (30, 50, 64, 96), (37, 41, 32, 64), (100, 80, 100, 32), (64, 64, 64, 64), (75, 33, 24, 24)]:
        img = rng.integers(0, 256, (Hin, Win, 3), dtype=np.uint8)
        np.testing.assert_array_equal(resample(img, Wout, Hout), np.asarray(Image.fromarray(img).resize((Wout, Hout), Image.BILINEAR)))


def test_console_scripts_are_registered_like_the_reference():
    """pyproject.toml registers `train` / `evaluate` (reference pyproject.toml:41-45) and the targets resolve to callables."""
    import importlib

    import tomli
    cfg = tomli.loads((ROOT / "pyproject.toml").read_text())
    scripts = cfg["project"]["scripts"]
    assert {"train", "evaluate"} <= set(scripts)
    for name, target in scripts.items():
        module, func = target.split(":")
        assert callable(getattr(importlib.import_module(module), func)), name
    assert "csrc/libsdnet_hip.so" in cfg["tool"]["setuptools"]["package-data"]["structuredetector_amd"]


def test_dispatch_options_are_thread_local():
    import ctypes as C
    """VERDICT r2 (boundary hygiene): the kernel-selection thresholds are per host thread -- a second thread sees the defaults while the
    first one has them changed, and its own changes do not leak back (two engines in one process cannot race on them)."""
    import threading

    from structuredetector_amd import _lib as L
    lib = L.lib()
    d = L.ConvDesc()
    d.B, d.Hi, d.Wi, d.Cin, d.Ho, d.Wo, d.Cout, d.R, d.S, d.stride, d.pad = 64, 64, 64, 128, 64, 64, 128, 3, 3, 1, 1
    name = lambda: lib.sd_conv2d_kernel_name(C.byref(d), 0).decode()
    default = name()
    assert default == "k_conv3x3_patch<128, false>"
    seen = {}
    gate_a, gate_b = threading.Event(), threading.Event()

    def other():
        gate_a.wait(10)
        seen["other_default"] = name()                                   # main thread has the patch kernel switched off right now
        assert lib.sd_set_option(b"conv_patch_min_tiles", 1 << 30) == 0   # ... and this thread switches it off for itself only
        seen["other_changed"] = name()
        gate_b.set()

    t = threading.Thread(target=other)
    t.start()
    try:
        assert lib.sd_set_option(b"conv_patch_min_tiles", 1 << 30) == 0
        changed = name()
        assert changed != default
        gate_a.set()
        assert gate_b.wait(10)
    finally:
        assert lib.sd_set_option(b"conv_patch_min_tiles", 512) == 0
        t.join(10)
    assert seen["other_default"] == default and seen["other_changed"] == changed
    assert name() == default                                             # the other thread's change never reached this one


def test_bench_self_launch_spawns_ranks_before_any_gpu_call(monkeypatch):
    """`python bench.py --gpus N` without a launcher (WORLD_SIZE unset) must not die on a WORLD_SIZE assertion after touching the GPU:
    it spawns `torch.distributed.run --nproc-per-node N bench.py ...` as a child and returns its exit code.  Checked on the command it
    builds (no process started) and on the launcher-mismatch refusal, which comes before the first torch.cuda call."""
    import importlib
    import subprocess
    import sys
    root = Path(__file__).resolve().parent.parent
    sys.path.insert(0, str(root))
    bench = importlib.import_module("bench")
    seen = {}

    def fake_run(cmd, env=None, **kw):
        seen["cmd"], seen["env"] = cmd, env
        return subprocess.CompletedProcess(cmd, 7)

    monkeypatch.setattr(subprocess, "run", fake_run)
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "8", "--steps", "3", "--warmup", "1"])
    import torch
    monkeypatch.setattr(torch.cuda, "is_available", lambda: pytest.fail("GPU touched before the ranks were spawned"))
    with pytest.raises(SystemExit) as e:
        bench.main()
    assert e.value.code == 7                                          # the child's exit code is relayed
    cmd = seen["cmd"]
    assert cmd[1:4] == ["-m", "torch.distributed.run", "--nnodes=1"] and "--nproc-per-node=8" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and int(cmd[cmd.index("--master-port") + 1]) > 0
    assert cmd[-6:] == ["--gpus", "8", "--steps", "3", "--warmup", "1"] and cmd[-7].endswith("bench.py")
    assert seen["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0" and seen["env"]["MASTER_ADDR"] == "127.0.0.1"
    # a launcher whose WORLD_SIZE disagrees with --gpus is refused before the first GPU call
    monkeypatch.setenv("WORLD_SIZE", "2")
    with pytest.raises(SystemExit, match="WORLD_SIZE=2"):
        bench.main()


def _torchvision_resnet34_state_dict(seed=0):
    """A state_dict in torchvision's resnet34 key schema (conv1 / bn1 / layer1..4 / fc; published BasicBlock [3, 4, 6, 3]), seeded values."""
    import torch
    g = torch.Generator().manual_seed(seed)
    sd = {}

    def bn(prefix, c):
        sd[f"{prefix}.weight"] = torch.rand(c, generator=g) + 0.5
        sd[f"{prefix}.bias"] = torch.randn(c, generator=g) * 0.1
        sd[f"{prefix}.running_mean"] = torch.randn(c, generator=g) * 0.1
        sd[f"{prefix}.running_var"] = torch.rand(c, generator=g) + 0.5
        sd[f"{prefix}.num_batches_tracked"] = torch.tensor(1234, dtype=torch.long)

    sd["conv1.weight"] = torch.randn(64, 3, 7, 7, generator=g) * 0.05
    bn("bn1", 64)
    cin = 64
    for li, (n, c) in enumerate(((3, 64), (4, 128), (6, 256), (3, 512)), start=1):
        for b in range(n):
            p = f"layer{li}.{b}"
            sd[f"{p}.conv1.weight"] = torch.randn(c, cin, 3, 3, generator=g) * 0.02
            bn(f"{p}.bn1", c)
            sd[f"{p}.conv2.weight"] = torch.randn(c, c, 3, 3, generator=g) * 0.02
            bn(f"{p}.bn2", c)
            if b == 0 and li > 1:
                sd[f"{p}.downsample.0.weight"] = torch.randn(c, cin, 1, 1, generator=g) * 0.05
                bn(f"{p}.downsample.1", c)
            cin = c
    sd["fc.weight"] = torch.randn(1000, 512, generator=g) * 0.01
    sd["fc.bias"] = torch.zeros(1000)
    return sd


def test_pretrained_backbone_is_loaded_from_a_local_torchvision_checkpoint(tmp_path, monkeypatch):
    """network.py:41-50: `pretrained=True` starts the trunk from torchvision's ImageNet ResNet-34 (conv1 / bn1 -> adpater.0 / .1,
    layerL -> downL, fc dropped).  Here the checkpoint is looked up locally (flag, environment, torch hub cache); a missing file is
    announced with a RuntimeWarning, a wrong one is refused."""
    import warnings

    import torch

    from structuredetector_amd import _lib as L
    from structuredetector_amd.model import network as PN
    sd = _torchvision_resnet34_state_dict()
    assert len(sd) == 216 + 2 and sum(v.numel() for k, v in sd.items() if v.dtype != torch.long and "running" not in k) == 21_797_672
    path = tmp_path / "resnet34-b627a593.pth"
    torch.save(sd, path)
    monkeypatch.delenv("SDNET_BACKBONE_WEIGHTS", raising=False)
    monkeypatch.setenv("TORCH_HOME", str(tmp_path / "nothing_here"))
    monkeypatch.setenv("XDG_CACHE_HOME", str(tmp_path / "nothing_here_either"))
    ns = lambda **kw: Namespace(labels={"a": 0, "b": 1}, parts={"p": 0}, fpn_depth=128, **kw)

    def check(net):
        own = net.state_dict()
        assert torch.equal(own["adpater.0.weight"], sd["conv1.weight"]) and torch.equal(own["adpater.1.running_var"], sd["bn1.running_var"])
        assert int(own["adpater.1.num_batches_tracked"]) == 1234
        for k, v in sd.items():
            if k.startswith("layer"):
                assert torch.equal(own["down" + k[5:]], v), k
        assert not any(k.startswith("fc.") for k in own)
        assert net.backbone_weights is not None

    # 1. explicit flag
    with warnings.catch_warnings():
        warnings.simplefilter("error")
        net = PN.Network(ns(backbone_weights=str(path)), pretrained=True)
    check(net)
    base = PN.Network(ns(), pretrained=False)                              # FPN / head keep their own initialisation
    for k in ("up1.weight", "up3.conv.0.weight", "head.conv.bias"):
        assert torch.equal(net.state_dict()[k], base.state_dict()[k])
    assert not torch.equal(net.state_dict()["down3.2.conv1.weight"], base.state_dict()["down3.2.conv1.weight"])
    # 2. environment variable, 3. torch hub cache under $TORCH_HOME
    monkeypatch.setenv("SDNET_BACKBONE_WEIGHTS", str(path))
    check(PN.Network(ns(), pretrained=True))
    monkeypatch.delenv("SDNET_BACKBONE_WEIGHTS")
    hub = tmp_path / "home" / "hub" / "checkpoints"
    hub.mkdir(parents=True)
    (hub / path.name).write_bytes(path.read_bytes())
    monkeypatch.setenv("TORCH_HOME", str(tmp_path / "home"))
    check(PN.Network(ns(), pretrained=True))
    # nothing found: loud warning, random trunk; pretrained=False: silent
    monkeypatch.setenv("TORCH_HOME", str(tmp_path / "nothing_here"))
    with pytest.warns(RuntimeWarning, match="RANDOM weights"):
        net = PN.Network(ns(), pretrained=True)
    assert net.backbone_weights is None
    with warnings.catch_warnings():
        warnings.simplefilter("error")
        PN.Network(ns(), pretrained=False)
    # a named file that does not exist, and a checkpoint of the wrong architecture, are errors
    with pytest.raises(L.SdError, match="not found"):
        PN.Network(ns(backbone_weights=str(tmp_path / "missing.pth")), pretrained=True)
    bad = dict(sd)
    del bad["layer3.5.conv2.weight"]
    bad["layer4.0.conv1.weight"] = torch.zeros(512, 256, 1, 1)
    torch.save(bad, tmp_path / "bad.pth")
    with pytest.raises(L.SdError, match="not a torchvision ResNet-34"):
        PN.Network(ns(backbone_weights=str(tmp_path / "bad.pth")), pretrained=True)


def test_small_utils_against_the_reference(golden_dir):
    """`AverageMeter`, `dict_grouping` and `draw_heatmaps` (src/sdnet/utils/utils.py:311-324,470-474, visualization.py:53-91) against values
    produced by the imported reference (tests/golden/gen_goldens.py::gen_small_utils): the last names of `sdnet.utils` this package lacked."""
    import torch
    from structuredetector_amd.utils import AverageMeter, dict_grouping, draw_heatmaps
    g = np.load(golden_dir / "small_utils.npz")
    m = AverageMeter()
    avgs = [m.update(float(v)) for v in g["meter_values"]]
    assert np.array_equal(np.array(avgs), g["meter_avgs"]) and m.sum == float(g["meter_sum"]) and m.count == int(g["meter_count"])
    m.reset()
    assert (m.sum, m.count, m.avg) == (0.0, 0, 0.0)
    got = dict_grouping([str(w) for w in g["words"]], key=len)
    want = json.loads(str(g["grouped"]))
    assert {str(k): v for k, v in got.items()} == want and [str(k) for k in got] == list(want)          # groups in first-seen order too
    labels = {f"label{i}": i for i in range(3)}; parts = {f"part{i}": i for i in range(2)}
    args = Namespace(_r_labels={v: k for k, v in labels.items()}, _r_parts={v: k for k, v in parts.items()},
                     _label_color_map={f"label{i}": tuple(int(c) for c in g["label_colors"][i]) for i in range(3)},
                     _part_color_map={f"part{i}": tuple(int(c) for c in g["part_colors"][i]) for i in range(2)})
    ca, cq = draw_heatmaps(torch.from_numpy(g["anchor_hm"]), torch.from_numpy(g["part_hm"]), args)
    assert ca.dtype == torch.uint8 and np.array_equal(ca.numpy(), g["anchor_rgb"]) and np.array_equal(cq.numpy(), g["part_rgb"])
    with pytest.raises(AssertionError):
        draw_heatmaps(torch.zeros(1, 3, 4, 4), torch.zeros(1, 2, 4, 4), args)


def test_debug_drawings_place_the_reference_primitives():
    """`draw_kp_and_emb` / `draw_embeddings` (visualization.py:94-170): discs of 1 % of the shorter side for peaks at or above the confidence
    threshold, a segment along every drawn part's embedding, the embedding field as red segments from every fourth output pixel -- compared
    with the same PIL primitives issued directly (the reference reaches them through torchvision's to_pil_image, absent here)."""
    import torch
    from PIL import Image, ImageDraw
    from structuredetector_amd.utils import draw_embeddings, draw_kp_and_emb, un_normalize
    args = Namespace(conf_threshold=0.5, down_ratio=4.0, _r_labels={0: "bean", 1: "maize"}, _r_parts={0: "leaf"},
                     _label_color_map={"bean": (200, 30, 30), "maize": (30, 200, 30)}, _part_color_map={"leaf": (30, 30, 200)})
    image = torch.zeros(3, 64, 96)
    base = Image.fromarray((un_normalize(image).clamp(0, 1) * 255).round().to(torch.uint8).permute(1, 2, 0).numpy())
    topk_obj = (torch.tensor([[0.9, 0.4]]), torch.tensor([[5, 9]]), torch.tensor([[1.0, 0.0]]), torch.tensor([[3.0, 10.0]]), torch.tensor([[4.0, 12.0]]))
    topk_kp = (torch.tensor([[0.7, 0.5, 0.2]]), torch.tensor([[1, 2, 3]]), torch.zeros(1, 3), torch.tensor([[8.0, 2.0, 6.0]]), torch.tensor([[20.0, 6.0, 1.0]]))
    emb = torch.tensor([[[-2.0, 1.5], [3.0, 0.0], [9.0, 9.0]]])
    got = draw_kp_and_emb(image, topk_obj, topk_kp, emb, args)
    want = base.copy(); pen = ImageDraw.Draw(want)
    pen.ellipse([16.0, 12.0, 16.0, 12.0], fill=(30, 200, 30), outline=(30, 200, 30))                     # radius int(64 * 0.01) = 0
    for (x, y, ex, ey) in ((80.0, 32.0, -2.0, 1.5), (24.0, 8.0, 3.0, 0.0)):                              # the 0.5 part is drawn (>= threshold), 0.2 is not
        pen.ellipse([x, y, x, y], fill=(30, 30, 200), outline=(30, 30, 200))
        pen.line([x, y, x + 4.0 * ex, y + 4.0 * ey], fill=(30, 30, 200), width=0)
    assert np.array_equal(np.asarray(got), np.asarray(want))
    field = torch.zeros(1, 2, 16, 24); field[0, 0, 4, 8] = 2.0; field[0, 1, 4, 8] = -1.0; field[0, 0, 5, 8] = 50.0   # (row 5 is not sampled)
    got = draw_embeddings(image, field, args)
    want = base.copy(); pen = ImageDraw.Draw(want)
    for j in range(0, 16, 4):
        for i in range(0, 24, 4):
            dx, dy = float(field[0, 0, j, i]) * 4.0, float(field[0, 1, j, i]) * 4.0
            pen.line([i * 4.0, j * 4.0, i * 4.0 + dx, j * 4.0 + dy], fill=(255, 0, 0), width=0)
    assert np.array_equal(np.asarray(got), np.asarray(want))
    with pytest.raises(AssertionError):
        draw_embeddings(image, torch.zeros(2, 2, 4, 4), args)



SAN_RUNTIME = Path("/opt/rocm/lib/llvm/lib/clang/22/lib/linux/libclang_rt.asan-x86_64.so")
SAN_TESTS = ["test_library_exports_every_declared_symbol", "test_c_abi_rejects_bad_arguments_without_touching_the_gpu",
             "test_set_option_is_host_only_and_rejects_unknown_names", "test_dispatch_options_are_thread_local",
             "test_roctx_ranges_behind_the_c_abi"]


def test_host_half_under_asan_ubsan():
    """SURVEY.md section 5 ("Race detection / sanitizers": none in the reference): `make SAN=1` builds libsdnet_hip_san.so with
    AddressSanitizer + UndefinedBehaviorSanitizer on the HOST half (-fno-sanitize-recover: the first finding aborts); a child python
    with the shared ASan runtime preloaded runs the tests that drive the validating entry points, every size query, the thread-local
    option tables and the roctx loader through it.  CPU container only (GPU sanitizers are unavailable on the pool)."""
    import glob
    runtime = SAN_RUNTIME if SAN_RUNTIME.exists() else next((Path(p) for p in glob.glob("/opt/rocm/lib/llvm/lib/clang/*/lib/linux/libclang_rt.asan-x86_64.so")), None)
    if runtime is None:
        pytest.skip("no shared ASan runtime in this image")
    if os.environ.get("SDNET_UNDER_SAN") == "1":
        pytest.skip("already inside the sanitizer child")
    csrc = ROOT / "structuredetector_amd" / "csrc"
    subprocess.run(["make", "-C", str(csrc), "SAN=1", "-j6"], check=True, capture_output=True, timeout=900)
    san = csrc / "libsdnet_hip_san.so"
    assert san.exists()
    env = dict(os.environ, LD_PRELOAD=str(runtime), ASAN_OPTIONS="detect_leaks=0:abort_on_error=0:exitcode=23",
               UBSAN_OPTIONS="print_stacktrace=1:halt_on_error=1", SDNET_HIP_LIB=str(san), SDNET_UNDER_SAN="1", SDNET_ROCTX="1")
    r = subprocess.run([sys.executable, "-m", "pytest", "-q", "-x", "-p", "no:cacheprovider", str(ROOT / "tests" / "test_host_cpu.py"), "-k",
                        " or ".join(SAN_TESTS)], cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    tail = (r.stdout + r.stderr)[-3000:]
    assert r.returncode == 0, tail
    assert f"{len(SAN_TESTS)} passed" in r.stdout, tail
    assert "AddressSanitizer" not in tail and "runtime error" not in tail, tail


def test_roctx_ranges_behind_the_c_abi():
    """sd_range_* (include/sdnet_hip.h; no reference counterpart): with SDNET_ROCTX=1 in the environment of the FIRST call the marker
    library is resolved and ranges nest; without it every call is a no-op.  Each case runs in its own child (the switch is read once)."""
    code = ("from structuredetector_amd import _lib as L\n"
            "from structuredetector_amd.utils import trace as T\n"
            "lib = L.lib()\n"
            "print(int(T.enabled()), lib.sd_range_library().decode())\n"
            "with T.span('step'):\n"
            "    with T.span('forward'):\n"
            "        T.mark('m')\n"
            "assert lib.sd_range_push(None) == (-1 if T.enabled() else 0)\n"
            "assert lib.sd_range_pop() == 0\n")
    for on in ("1", "0"):
        env = {k: v for k, v in os.environ.items() if k != "SDNET_ROCTX"}
        if on == "1":
            env["SDNET_ROCTX"] = "1"
        r = subprocess.run([sys.executable, "-c", code], cwd=ROOT, env=env, capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, r.stderr[-2000:]
        flag, name = (r.stdout.split() + [""])[:2]
        if on == "1":
            assert flag == "1" and "roctx" in name, r.stdout           # the image ships librocprofiler-sdk-roctx / libroctx64
        else:
            assert flag == "0" and name == ""


def test_rows16_kernel_isa_has_no_unseen_mfma_hazard():
    """k_conv3x3_c64_rows16_bf16's MFMAs are inline asm, invisible to hipcc's hazard recogniser: safe only while the register allocator keeps
    every weight fragment where its constraint pinned it (a parked fragment is copied into an AGPR quad in front of the MFMA without the wait
    states in between -- seen in round 5 as wrong, run-to-run different sums).  tools/check_rows16_isa.py compiles the translation unit to ISA
    and checks every DISPATCHED instantiation: no weight AGPR written, nothing spilled, inside the row loops."""
    r = subprocess.run([sys.executable, str(ROOT / "tools" / "check_rows16_isa.py")], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert "ok" in r.stdout
