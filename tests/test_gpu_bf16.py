"""bf16 backbone (inference): hand-written bf16 MFMA convs against a PyTorch reference of the same op
(bf16-rounded operands, fp32 accumulation) and the whole network against the CPU oracle under bf16 autocast."""
import ctypes as C
from argparse import Namespace

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import sdnet_oracle as O
from tests.test_gpu_network import close, make_desc

pytestmark = pytest.mark.gpu
DEV = "cuda"


def nhwc_bf16(t):
    return t.permute(0, 2, 3, 1).contiguous().to(DEV).to(torch.bfloat16)


@pytest.mark.parametrize("case", [(2, 16, 24, 64, 64, 3, 1, 1), (1, 20, 12, 64, 128, 3, 2, 1), (2, 12, 12, 128, 128, 1, 1, 0),
                                  (1, 16, 16, 512, 512, 3, 1, 1), (3, 9, 7, 256, 128, 3, 1, 1)])
def test_conv_fwd_bf16(case):
    from structuredetector_amd import _lib as L
    B, H, W, cin, cout, k, stride, pad = case
    g = torch.Generator().manual_seed(sum(case))
    x = torch.randn(B, cin, H, W, generator=g).bfloat16().float()          # exactly representable operands
    w = (torch.randn(cout, cin, k, k, generator=g) / (cin * k * k) ** 0.5).bfloat16().float()
    scale = torch.rand(cout, generator=g) + 0.5; shift = torch.randn(cout, generator=g)
    lib = L.lib()
    d = make_desc(L, B, H, W, cin, cout, k, stride, pad)
    res = torch.randn(B, cout, d.Ho, d.Wo, generator=g).bfloat16().float()
    xd, wd, rd = nhwc_bf16(x), nhwc_bf16(w), nhwc_bf16(res)
    sc, sh = scale.to(DEV), shift.to(DEV)
    y = torch.empty(B, d.Ho, d.Wo, cout, dtype=torch.bfloat16, device=DEV)
    nws = lib.sd_conv2d_fwd_bf16_workspace_bytes(C.byref(d))
    ws = torch.empty(max(nws, 256), dtype=torch.uint8, device=DEV)
    L.check(lib.sd_conv2d_fwd_bf16(xd.data_ptr(), wd.data_ptr(), y.data_ptr(), C.byref(d), sc.data_ptr(), sh.data_ptr(), rd.data_ptr(), 0, 1,
                                   ws.data_ptr(), ws.numel(), L.stream()))
    ref = F.relu(F.conv2d(x, w, None, stride, pad) * scale[None, :, None, None] + shift[None, :, None, None] + res)
    got = y.float().permute(0, 3, 1, 2).cpu()
    # products of bf16 values are exact in fp32; only the accumulation order and the final bf16 rounding (2^-9) differ
    close(got, ref, 6e-3)


@pytest.mark.parametrize("case", [(3, 16, 16, 64, 64), (2, 32, 32, 128, 128), (1, 64, 64, 64, 128), (1, 8, 128, 64, 64), (1, 4, 128, 256, 128)])
def test_conv3x3_patch_kernel_bf16(case):
    """bf16 instantiation of k_conv3x3_patch (forced on for small grids): fused affine + residual + ReLU epilogue, double-buffered
    and rolling patches, 64- and 128-channel tiles."""
    from structuredetector_amd import _lib as L
    B, H, W, cin, cout = case
    g = torch.Generator().manual_seed(sum(case) + 1)
    x = torch.randn(B, cin, H, W, generator=g).bfloat16().float()
    w = (torch.randn(cout, cin, 3, 3, generator=g) / (cin * 9) ** 0.5).bfloat16().float()
    scale = torch.rand(cout, generator=g) + 0.5; shift = torch.randn(cout, generator=g)
    res = torch.randn(B, cout, H, W, generator=g).bfloat16().float()
    lib = L.lib()
    d = make_desc(L, B, H, W, cin, cout, 3, 1, 1)
    xd, wd, rd = nhwc_bf16(x), nhwc_bf16(w), nhwc_bf16(res)
    sc, sh = scale.to(DEV), shift.to(DEV)
    y = torch.empty(B, H, W, cout, dtype=torch.bfloat16, device=DEV)
    L.check(lib.sd_set_option(b"conv_patch_min_tiles", 1))
    L.check(lib.sd_set_option(b"conv_patch_bn64", 1))
    try:
        # no workspace -> no split-K -> the patch kernel takes the launch
        L.check(lib.sd_conv2d_fwd_bf16(xd.data_ptr(), wd.data_ptr(), y.data_ptr(), C.byref(d), sc.data_ptr(), sh.data_ptr(), rd.data_ptr(), 0, 1,
                                       0, 0, L.stream()))
    finally:
        L.check(lib.sd_set_option(b"conv_patch_min_tiles", 512))
        L.check(lib.sd_set_option(b"conv_patch_bn64", 0))
    ref = F.relu(F.conv2d(x, w, None, 1, 1) * scale[None, :, None, None] + shift[None, :, None, None] + res)
    close(y.float().permute(0, 3, 1, 2).cpu(), ref, 6e-3)


def test_stem_conv_on_the_bf16_mfma():
    """sd_conv2d_stem_fwd(out_bf16=1): image and weights rounded to bf16, fp32 accumulation, folded affine + ReLU, bf16 NHWC output."""
    from structuredetector_amd import _lib as L
    lib = L.lib()
    for (B, H, W) in ((2, 64, 96), (1, 32, 320)):
        g = torch.Generator().manual_seed(H * 3 + W)
        x = torch.randn(B, 3, H, W, generator=g)
        w = torch.randn(64, 3, 7, 7, generator=g) / 147 ** 0.5
        scale = torch.rand(64, generator=g) + 0.5; shift = torch.randn(64, generator=g) * 0.2
        d = make_desc(L, B, H, W, 3, 64, 7, 2, 3)
        y = torch.empty(B, d.Ho, d.Wo, 64, dtype=torch.bfloat16, device=DEV)
        ws = torch.empty(lib.sd_conv2d_stem_fwd_workspace_bytes(C.byref(d)), dtype=torch.uint8, device=DEV)
        xd, wd = x.to(DEV), w.permute(0, 2, 3, 1).contiguous().to(DEV)
        sc, sh = scale.to(DEV), shift.to(DEV)
        L.check(lib.sd_conv2d_stem_fwd(xd.data_ptr(), wd.data_ptr(), y.data_ptr(), C.byref(d), sc.data_ptr(), sh.data_ptr(), 1, 1, ws.data_ptr(),
                                       ws.numel(), L.stream()))
        ref = F.relu(F.conv2d(x.bfloat16().float(), w.bfloat16().float(), None, 2, 3) * scale[None, :, None, None] + shift[None, :, None, None])
        close(y.float().permute(0, 3, 1, 2).cpu(), ref, 6e-3)


def _pair(M=2, N=1, seed=0):
    from structuredetector_amd.model import Network
    ref = O.build_reference_network(M, N, seed=seed)
    args = Namespace(labels={f"l{i}": i for i in range(M)}, parts={f"p{i}": i for i in range(N)}, fpn_depth=128, use_amp=True)
    net = Network(args, pretrained=False, raw_output=True)
    net.load_state_dict(ref.state_dict())
    return ref, net.to(DEV)


@pytest.mark.parametrize("B,H,W", [(2, 128, 160), (3, 96, 160), (5, 64, 64), (1, 224, 96), (7, 128, 32)])
def test_network_bf16_inference_vs_oracle(B, H, W):
    """bf16 backbone, eval forward, at even and odd batch sizes and non-square inputs (down to a 1-pixel-high layer4 map)."""
    ref, net = _pair(seed=2)
    x = torch.randn(B, 3, H, W, generator=torch.Generator().manual_seed(1))
    ref.eval(); net.eval()
    with torch.no_grad():
        want32 = ref(x)
        with torch.autocast(device_type="cpu", dtype=torch.bfloat16):
            want16 = ref(x).float()
        got = net(x.to(DEV)).cpu()
    assert got.dtype == torch.float32 and got.shape == want32.shape
    scale = want32.abs().max().item()
    err_ours = (got - want32).abs().max().item() / scale
    err_autocast = (want16 - want32).abs().max().item() / scale
    # fp32 epilogues make the hand-written path at least as accurate as torch's bf16 autocast of the same network
    assert err_ours <= max(1.5 * err_autocast, 2e-2), (err_ours, err_autocast)
    # fp32 path of the same object still available and exact
    net.bf16_inference = False
    with torch.no_grad():
        close(net(x.to(DEV)).cpu(), want32, 1e-4)


@pytest.mark.parametrize("M,N,B,H,W", [(2, 1, 2, 128, 256), (8, 8, 1, 256, 256), (2, 1, 3, 512, 128)])
def test_head_fused_into_the_last_fpn_conv(M, N, B, H, W):
    """sd_conv2d_fwd_bf16_head (network.py:17-18 + 22-29 in one launch: the 1x1 head applied to every tile of `up4.conv` in the two-group
    kernel's epilogue, the FPN output never stored) against the same network with the head as its own launch: 7 and 20 head channels, maps
    of 32 .. 128 pixels (whole rows and column strips); then against the oracle like every other bf16 forward."""
    from structuredetector_amd import _lib as L
    lib = L.lib()
    ref, net = _pair(M, N, seed=4)
    x = torch.randn(B, 3, H, W, generator=torch.Generator().manual_seed(2))
    ref.eval(); net.eval()
    L.check(lib.sd_set_option(b"conv_pp_min_tiles", 1))
    L.check(lib.sd_set_option(b"conv_fwd_split_k", 0))
    try:
        d = make_desc(L, B, H // 4, W // 4, 128, 128, 3, 1, 1)
        assert lib.sd_conv2d_fwd_bf16_head_supported(C.byref(d), M + N + 4) == 1
        net._engine.small_batch_kernel = False        # (small batches: the plain path would take sd_conv2d_fwd_sb, another summation order)
        with torch.no_grad():
            net._engine.fuse_head = True
            fused = net(x.to(DEV)).cpu()
            net._engine.fuse_head = False
            plain = net(x.to(DEV)).cpu()
            want32 = ref(x)
            with torch.autocast(device_type="cpu", dtype=torch.bfloat16):
                want16 = ref(x).float()
    finally:
        net._engine.fuse_head = True
        net._engine.small_batch_kernel = True
        L.check(lib.sd_set_option(b"conv_pp_min_tiles", 200))
        L.check(lib.sd_set_option(b"conv_fwd_split_k", 1))
    assert fused.shape == plain.shape == want32.shape
    # same bf16 FPN output, same hi + lo split of the head weights; the two kernels sum the 128 products in different orders
    scale = plain.abs().max().item()
    assert (fused - plain).abs().max().item() <= 2e-5 * scale, ((fused - plain).abs().max().item(), scale)
    s32 = want32.abs().max().item()
    assert (fused - want32).abs().max().item() / s32 <= max(1.5 * (want16 - want32).abs().max().item() / s32, 2e-2)


def test_graphed_bf16_forward_is_the_eager_bf16_forward():
    """Network.graphed captures the forward `net(x)` runs: with `bf16_inference` the bf16 backbone (round 4: it captured the fp32 path) --
    the replay reproduces the eager bf16 forward bit for bit, also on a second input written into the graph's static buffer."""
    ref, net = _pair(seed=6)
    net.eval()
    g = torch.Generator().manual_seed(3)
    x1, x2 = (torch.randn(1, 3, 256, 256, generator=g).to(DEV) for _ in range(2))
    with torch.no_grad():
        e1, e2 = net(x1).clone(), net(x2).clone()
        run = net.graphed(x1)
        assert torch.equal(run(x1), e1)
        run.static_in.copy_(x2)
        assert torch.equal(run(run.static_in), e2)
        net.bf16_inference = False
        f1 = net(x1)
    assert not torch.equal(f1, e1)                    # (the fp32 forward is a different tensor: the replay above was not it)


def test_stress_config_bf16_backbone_fp32_decode():
    """BASELINE configs[4]: 1024x1024, 8 labels / 8 parts, >= 64 objects per image, bf16 backbone + fp32 decode.
    Decoder parity at this size is asserted against the oracle on the SAME head tensor (bit-exact indices / grouping)."""
    from structuredetector_amd.data import Decoder
    from structuredetector_amd.model import Network
    from tests.test_host_cpu import make_args
    M = N = 8; K, P = 128, 512
    args = make_args(M, N, K, P, device=torch.device(DEV), use_amp=True)
    net = Network(args, pretrained=False).to(DEV).eval()
    x = torch.randn(1, 3, 1024, 1024, device=DEV, generator=torch.Generator(DEV).manual_seed(3))
    with torch.no_grad():
        out = net(x)
    assert out["anchor_hm"].shape == (1, 8, 256, 256) and out["anchor_hm"].dtype == torch.float32
    head = torch.cat([out["anchor_hm"], out["part_hm"], out["offsets"], out["embeddings"]], 1)
    assert torch.isfinite(head).all()
    dec = Decoder(args)
    packed, _ = dec.decode_packed(out, 0.5, 0.1)
    got = dec.split_packed(packed.cpu().numpy(), 1, K, P)
    h = head.cpu().numpy()
    t = O.decode_tensors(h[:, :M], h[:, M:M + N], h[:, M + N:M + N + 2], h[:, M + N + 2:], K, P, 0.5, 0.1)
    # indices on the safe ranks AND the grouping of every safe part, unconditionally (tests/helpers.py); the dense
    # >= 64-object workload of this config is asserted in test_gpu_parity.py::test_decoder_dense_stress_scenes_grouping_vs_oracle
    from tests.helpers import assert_decode_matches_oracle
    # (a random-init network saturates many logits: clamp plateaus leave few tie-free ranks here, so no coverage floor)
    assert_decode_matches_oracle(got, t, 0.5, dict(rtol=4e-7, atol=0))


def test_network_bf16_bs1_512_small_batch_kernel_vs_autocast_oracle():
    """BASELINE configs[1] on the bf16 backbone: batch 1, 512x512 -- every conv on sd_conv2d_fwd_sb (bf16 MFMA, K split over blocks and
    combined inside the launch) -- against the oracle's fp32 forward, with torch.autocast(cpu, bfloat16) of the same oracle as the
    yardstick; also the same image inside a batch of 3 (other tile / slice decomposition) and bit-reproducibility."""
    ref, net = _pair(seed=5)
    g = torch.Generator().manual_seed(8)
    x = torch.randn(3, 3, 512, 512, generator=g)
    ref.eval(); net.eval()
    with torch.no_grad():
        want32 = ref(x)
        with torch.autocast(device_type="cpu", dtype=torch.bfloat16):
            want16 = ref(x).float()
        one = net(x[:1].to(DEV)).cpu()
        again = net(x[:1].to(DEV)).cpu()
        three = net(x.to(DEV)).cpu()
    assert torch.equal(one, again)
    scale = want32.abs().max().item()
    err_auto = (want16 - want32).abs().max().item() / scale
    for got, want in ((one, want32[:1]), (three, want32)):
        err = (got - want).abs().max().item() / scale
        assert err <= max(1.5 * err_auto, 2e-2), (err, err_auto)
