"""Mixed-precision training (`--amp`, reference src/sdnet/model/trainer.py:115-121): bf16 activations / conv weights with fp32
accumulation, statistics, loss and master weights.  Kernel level: against PyTorch references of the same op on bf16-rounded
operands; whole step: against the CPU oracle under torch.autocast(cpu, bfloat16) with an fp64 run as the yardstick."""
import copy
import ctypes as C
import json

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import sdnet_oracle as O
from tests.test_gpu_network import close, make_desc

pytestmark = pytest.mark.gpu
DEV = "cuda"


def nhwc16(t):
    return t.permute(0, 2, 3, 1).contiguous().to(DEV).to(torch.bfloat16)


def back(t):
    return t.float().permute(0, 3, 1, 2).cpu()


@pytest.mark.parametrize("case", [(4, 16, 24, 64, 64, 3, 1, 1), (2, 20, 12, 64, 128, 3, 2, 1), (4, 12, 12, 64, 128, 1, 2, 0), (2, 16, 16, 512, 512, 3, 1, 1),
                                  (64, 32, 32, 128, 128, 3, 1, 1), (16, 64, 64, 64, 128, 3, 2, 1), (2, 6, 10, 512, 128, 1, 1, 0)])
def test_conv_bf16_forward_statistics_and_data_gradient(case):
    """sd_conv2d_fwd_bf16_bn_stats (output bf16 + BatchNorm batch statistics of the ROUNDED output, fused or via the small-batch
    path) and sd_conv2d_dgrad_bf16 (stride 1 / stride-2 parity classes / strided 1x1, plain, + residual, + half-size residual)."""
    from structuredetector_amd import _lib as L
    B, H, W, cin, cout, k, stride, pad = case
    g = torch.Generator().manual_seed(sum(case))
    x = torch.randn(B, cin, H, W, generator=g).bfloat16().float()
    w = (torch.randn(cout, cin, k, k, generator=g) / (cin * k * k) ** 0.5).bfloat16().float()
    lib = L.lib()
    d = make_desc(L, B, H, W, cin, cout, k, stride, pad)
    xd, wd = nhwc16(x), nhwc16(w)
    y = torch.empty(B, d.Ho, d.Wo, cout, dtype=torch.bfloat16, device=DEV)
    mean = torch.empty(cout, device=DEV); invstd = torch.empty(cout, device=DEV)
    rm = torch.zeros(cout, device=DEV); rv = torch.ones(cout, device=DEV)
    ws = torch.empty(lib.sd_conv2d_fwd_bf16_bn_stats_workspace_bytes(C.byref(d)), dtype=torch.uint8, device=DEV)
    L.check(lib.sd_conv2d_fwd_bf16_bn_stats(xd.data_ptr(), wd.data_ptr(), y.data_ptr(), C.byref(d), 1e-5, 0.1, rm.data_ptr(), rv.data_ptr(),
                                            mean.data_ptr(), invstd.data_ptr(), ws.data_ptr(), ws.numel(), L.stream()))
    ref = F.conv2d(x, w, None, stride, pad)
    close(back(y), ref, 6e-3)
    yr = back(y).double()                                            # statistics of the tensor that was stored
    m_ref = yr.mean((0, 2, 3)); v_ref = yr.var((0, 2, 3), unbiased=False)
    np.testing.assert_allclose(mean.cpu().numpy(), m_ref.numpy(), rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(invstd.cpu().numpy(), (1.0 / torch.sqrt(v_ref + 1e-5)).numpy(), rtol=1e-4)
    n = yr.numel() / cout
    np.testing.assert_allclose(rv.cpu().numpy(), (0.9 + 0.1 * v_ref * n / (n - 1)).numpy(), rtol=1e-4)
    # data gradient
    dy = torch.randn(B, cout, d.Ho, d.Wo, generator=g).bfloat16().float()
    wt = w.permute(1, 2, 3, 0).contiguous().to(DEV).to(torch.bfloat16)       # [Cin][R][S][Cout]
    dyd = nhwc16(dy)
    xg = x.clone().requires_grad_(True)
    F.conv2d(xg, w, None, stride, pad).backward(dy)
    dx_ref = xg.grad
    dx = torch.empty(B, H, W, cin, dtype=torch.bfloat16, device=DEV)
    L.check(lib.sd_conv2d_dgrad_bf16(dyd.data_ptr(), wt.data_ptr(), dx.data_ptr(), C.byref(d), 0, 0, L.stream()))
    close(back(dx), dx_ref, 8e-3)
    res = torch.randn(B, cin, H, W, generator=g).bfloat16().float()
    resd = nhwc16(res)
    L.check(lib.sd_conv2d_dgrad_bf16(dyd.data_ptr(), wt.data_ptr(), dx.data_ptr(), C.byref(d), resd.data_ptr(), 1, L.stream()))
    close(back(dx), dx_ref + res, 8e-3)
    if H % 2 == 0 and W % 2 == 0:
        half = torch.randn(B, cin, H // 2, W // 2, generator=g).bfloat16().float()
        full = torch.zeros(B, cin, H, W); full[:, :, ::2, ::2] = half
        half16 = nhwc16(half)
        L.check(lib.sd_conv2d_dgrad_bf16(dyd.data_ptr(), wt.data_ptr(), dx.data_ptr(), C.byref(d), half16.data_ptr(), 2, L.stream()))
        close(back(dx), dx_ref + full, 8e-3)


@pytest.mark.parametrize("case", [(2, 32, 32, 64, 64, 3, 1, 1), (1, 16, 16, 128, 64, 3, 1, 1), (64, 128, 128, 64, 64, 3, 1, 1), (8, 64, 64, 128, 128, 3, 1, 1),
                                  (4, 16, 16, 512, 512, 3, 1, 1), (3, 32, 64, 64, 128, 3, 1, 1), (2, 20, 12, 64, 128, 3, 2, 1), (4, 12, 12, 64, 128, 1, 2, 0),
                                  (2, 8, 8, 128, 128, 1, 1, 0), (2, 16, 16, 256, 512, 3, 2, 1), (3, 16, 16, 512, 128, 1, 1, 0), (1, 17, 9, 128, 256, 1, 2, 0),
                                  # the row-ring kernel: several (image, strip) segments per split, segments cut inside a strip, one-row maps
                                  (20, 4, 32, 64, 64, 3, 1, 1), (5, 12, 32, 64, 128, 3, 1, 1), (3, 7, 96, 128, 64, 3, 1, 1), (9, 1, 64, 64, 64, 3, 1, 1)])
def test_conv_bf16_weight_gradient(case):
    """sd_conv2d_wgrad_bf16: dW (fp32) from bf16 dy / x.  3x3 stride-1 layers take k_wgrad3x3_bf16 (bf16 MFMA fed by
    ds_read_b64_tr_b16 transposed reads; 32-wide rows and the two-rows-of-16 form; one and several pixel splits; image borders),
    strided 3x3 and 1x1 convs k_wgrad_tap_bf16 (one tap per block, 128 x 64 / 128 x 128 tiles, ragged last chunk).  Products of bf16 values are exact in fp32 and the accumulation
    is fp32 on both sides, so only the summation order differs from the reference."""
    from structuredetector_amd import _lib as L
    B, H, W, cin, cout, k, stride, pad = case
    g = torch.Generator().manual_seed(sum(case) + 7)
    x = torch.randn(B, cin, H, W, generator=g).bfloat16().float()
    lib = L.lib()
    d = make_desc(L, B, H, W, cin, cout, k, stride, pad)
    dy = torch.randn(B, cout, d.Ho, d.Wo, generator=g).bfloat16().float()
    wz = torch.zeros(cout, cin, k, k, requires_grad=True)
    xd, dyd = x.to(DEV), dy.to(DEV)
    wg = wz.detach().to(DEV).requires_grad_(True)
    F.conv2d(xd, wg, None, stride, pad).backward(dyd)                  # reference on the GPU (fp32, same operands)
    ref = wg.grad.cpu()
    dw = torch.full((cout, k, k, cin), float("nan"), device=DEV)
    ws = torch.empty(lib.sd_conv2d_wgrad_bf16_workspace_bytes(C.byref(d)), dtype=torch.uint8, device=DEV)
    dy16, x16 = nhwc16(dy), nhwc16(x)                                   # (named: raw pointers of temporaries would alias)
    L.check(lib.sd_conv2d_wgrad_bf16(dy16.data_ptr(), x16.data_ptr(), dw.data_ptr(), C.byref(d), 0, ws.data_ptr(), ws.numel(), L.stream()))
    close(dw.permute(0, 3, 1, 2).cpu(), ref, 2e-4)
    base = torch.randn(cout, k, k, cin, device=DEV, generator=torch.Generator(DEV).manual_seed(1))
    acc = base.clone()
    L.check(lib.sd_conv2d_wgrad_bf16(dy16.data_ptr(), x16.data_ptr(), acc.data_ptr(), C.byref(d), 1, ws.data_ptr(), ws.numel(), L.stream()))
    close((acc - base).permute(0, 3, 1, 2).cpu(), ref, 2e-4)            # accumulate = 1 adds into dW
    if k == 3 and stride == 1 and d.Wo % 32 == 0:
        # the other forms on the same operands (default 5 = two groups of four waves half a chunk apart in one 512-thread block, their sums
        # combined through LDS): 0 = one 3 x 34 patch per chunk, 2 .. 4 = one group per block, prefetch distance 2 .. 4
        for form in (0, 2, 3, 4):
            L.check(lib.sd_set_option(b"wgrad_bf16_ring", form))
            try:
                dw1 = torch.full_like(dw, float("nan"))
                L.check(lib.sd_conv2d_wgrad_bf16(dy16.data_ptr(), x16.data_ptr(), dw1.data_ptr(), C.byref(d), 0, ws.data_ptr(), ws.numel(), L.stream()))
            finally:
                L.check(lib.sd_set_option(b"wgrad_bf16_ring", 5))
            close(dw1.permute(0, 3, 1, 2).cpu(), ref, 2e-4)


def test_bn_kernels_on_bf16_activations_equal_the_fp32_kernels_rounded():
    """sd_bn_apply_bf16 / sd_bn_bwd_bf16 / sd_col_sum_bf16 / sd_upsample2x_bwd_bf16 / casts: the same arithmetic as the fp32 kernels
    on the widened operands, ONE rounding at the store -> bit-identical to round_bf16(fp32 kernel(float(x)))."""
    from structuredetector_amd import _lib as L
    lib = L.lib()
    g = torch.Generator(DEV).manual_seed(3)
    M, Cc = 4 * 24 * 16, 128
    x16 = torch.randn(M, Cc, device=DEV, generator=g).bfloat16()
    res16 = torch.randn(M, Cc, device=DEV, generator=g).bfloat16()
    dy16 = torch.randn(M, Cc, device=DEV, generator=g).bfloat16()
    gamma = torch.rand(Cc, device=DEV, generator=g) + 0.5; beta = torch.randn(Cc, device=DEV, generator=g) * 0.3
    mean = torch.randn(Cc, device=DEV, generator=g) * 0.1; invstd = torch.rand(Cc, device=DEV, generator=g) + 0.7
    # casts round-trip
    f = torch.empty(M, Cc, device=DEV)
    L.check(lib.sd_cast_bf16_to_f32(x16.data_ptr(), f.data_ptr(), f.numel(), L.stream()))
    assert torch.equal(f, x16.float())
    b = torch.empty(M, Cc, dtype=torch.bfloat16, device=DEV)
    r32 = torch.randn(M, Cc, device=DEV, generator=g)
    L.check(lib.sd_cast_f32_to_bf16(r32.data_ptr(), b.data_ptr(), b.numel(), L.stream()))
    assert torch.equal(b, r32.bfloat16())
    for relu, with_res, want_mask in ((1, False, False), (1, True, True), (0, False, False)):
        y16 = torch.empty_like(x16); y32 = torch.empty(M, Cc, device=DEV)
        m16 = torch.empty(M * Cc // 4, dtype=torch.uint8, device=DEV); m32 = torch.empty_like(m16)
        x32, res32 = x16.float(), res16.float()
        L.check(lib.sd_bn_apply_bf16(x16.data_ptr(), y16.data_ptr(), M, Cc, mean.data_ptr(), invstd.data_ptr(), gamma.data_ptr(), beta.data_ptr(),
                                     res16.data_ptr() if with_res else 0, relu, m16.data_ptr() if want_mask else 0, L.stream()))
        L.check(lib.sd_bn_apply(x32.data_ptr(), y32.data_ptr(), M, Cc, mean.data_ptr(), invstd.data_ptr(), gamma.data_ptr(), beta.data_ptr(),
                                res32.data_ptr() if with_res else 0, relu, m32.data_ptr() if want_mask else 0, L.stream()))
        assert torch.equal(y16, y32.bfloat16())
        if want_mask:
            assert torch.equal(m16, m32)
        # backward: relu mode 2 (mask recomputed from x) for plain layers, 3 (mask bytes) for residual layers, 0 for none
        mode = 0 if relu == 0 else (3 if want_mask else 2)
        wsb = torch.empty(lib.sd_col_reduce_workspace_bytes(M, Cc), dtype=torch.uint8, device=DEV)
        dx16 = torch.empty_like(x16); g16 = torch.empty_like(x16); dx32 = torch.empty(M, Cc, device=DEV); g32 = torch.empty(M, Cc, device=DEV)
        dg16 = torch.empty(Cc, device=DEV); db16 = torch.empty(Cc, device=DEV); dg32 = torch.empty(Cc, device=DEV); db32 = torch.empty(Cc, device=DEV)
        dy32 = dy16.float()
        L.check(lib.sd_bn_bwd_bf16(dy16.data_ptr(), x16.data_ptr(), m16.data_ptr() if mode == 3 else 0, mode, M, Cc, mean.data_ptr(), invstd.data_ptr(),
                                   gamma.data_ptr(), beta.data_ptr(), dx16.data_ptr(), g16.data_ptr(), dg16.data_ptr(), db16.data_ptr(), 0,
                                   wsb.data_ptr(), wsb.numel(), L.stream()))
        L.check(lib.sd_bn_bwd(dy32.data_ptr(), x32.data_ptr(), m32.data_ptr() if mode == 3 else 0, mode, M, Cc, mean.data_ptr(), invstd.data_ptr(),
                              gamma.data_ptr(), beta.data_ptr(), dx32.data_ptr(), g32.data_ptr(), dg32.data_ptr(), db32.data_ptr(), 0,
                              wsb.data_ptr(), wsb.numel(), L.stream()))
        # (the 16-byte bf16 reduction walks the rows in another order than the fp32 kernel: the per-channel sums agree to fp32
        #  rounding, dx to one bf16 ulp where a mean moved in its last bit; the masked gradient g is elementwise -> exact)
        assert torch.equal(g16, g32.bfloat16())
        np.testing.assert_allclose(dg16.cpu().numpy(), dg32.cpu().numpy(), rtol=2e-5, atol=1e-4)
        np.testing.assert_allclose(db16.cpu().numpy(), db32.cpu().numpy(), rtol=2e-5, atol=1e-4)
        diff = (dx16.float() - dx32.bfloat16().float()).abs()
        assert (diff <= dx32.abs() * 2 ** -7 + 1e-6).all() and (diff > 0).float().mean().item() < 0.01
    # bias-gradient column sums and the upsample backward
    wsb = torch.empty(lib.sd_col_reduce_workspace_bytes(M, Cc), dtype=torch.uint8, device=DEV)
    s16 = torch.empty(Cc, device=DEV); s32 = torch.empty(Cc, device=DEV)
    L.check(lib.sd_col_sum_bf16(dy16.data_ptr(), M, Cc, s16.data_ptr(), 0, wsb.data_ptr(), wsb.numel(), L.stream()))
    x32 = dy16.float()
    L.check(lib.sd_col_sum(x32.data_ptr(), M, Cc, s32.data_ptr(), 0, wsb.data_ptr(), wsb.numel(), L.stream()))
    assert torch.equal(s16, s32)
    up = torch.randn(2, 16, 24, 64, device=DEV, generator=g).bfloat16()
    o16 = torch.empty(2, 8, 12, 64, dtype=torch.bfloat16, device=DEV); o32 = torch.empty(2, 8, 12, 64, device=DEV)
    up32 = up.float()
    L.check(lib.sd_upsample2x_bwd_bf16(up.data_ptr(), 0, o16.data_ptr(), 2, 8, 12, 64, L.stream()))
    L.check(lib.sd_upsample2x_bwd(up32.data_ptr(), 0, o32.data_ptr(), 2, 8, 12, 64, L.stream()))
    assert torch.equal(o16, o32.bfloat16())


def _pair(seed=0):
    from argparse import Namespace
    from structuredetector_amd.model import Network
    ref = O.build_reference_network(2, 1, seed=seed)
    args = Namespace(labels={"l0": 0, "l1": 1}, parts={"p0": 0}, fpn_depth=128)
    net = Network(args, pretrained=False, raw_output=True)
    net.load_state_dict(ref.state_dict())
    return ref, net.to(DEV)


def test_amp_training_step_vs_oracle_under_autocast():
    """Forward + backward of the mixed-precision step against the oracle network under torch.autocast(cpu, bfloat16) -- the
    reference's own `--amp` mechanism (trainer.py:40-42,115-121).  bf16 rounding makes the two implementations differ by far more
    than 1e-4 from each other, so the yardstick is an fp64 run of the same network: the HIP step must be as close to the fp64 truth
    as the autocast oracle is (error populations over all parameter tensors within a factor 2)."""
    ref, net = _pair(seed=13)
    g = torch.Generator().manual_seed(5)
    x = torch.randn(8, 3, 128, 128, generator=g)
    dy = torch.randn(8, 7, 32, 32, generator=g) * 0.1
    ref.train(); net.train()
    ref64 = copy.deepcopy(ref).double()
    ref_ac = copy.deepcopy(ref)
    out64 = ref64(x.double()); out64.backward(dy.double())
    with torch.autocast(device_type="cpu", dtype=torch.bfloat16):
        out_ac = ref_ac(x)
    out_ac.float().backward(dy)
    out, tape = net.forward_train(x.to(DEV), amp=True)
    assert out.dtype == torch.float32 and tape["amp"] is True and tape["f1"].dtype == torch.bfloat16
    net.backward_from(tape, dy.to(DEV))
    scale = out64.abs().max().item()
    e_fwd_gpu = (out.cpu().double() - out64.detach()).abs().max().item() / scale
    e_fwd_ac = (out_ac.detach().double() - out64.detach()).abs().max().item() / scale
    assert e_fwd_gpu <= 2 * e_fwd_ac + 1e-3, (e_fwd_gpu, e_fwd_ac)
    g64, gac = dict(ref64.named_parameters()), dict(ref_ac.named_parameters())
    e_gpu, e_ac = [], []
    for name, p in net.named_parameters():
        truth = g64[name].grad
        s = truth.abs().max().item() + 1e-30
        e_gpu.append((net.grad_of(p).cpu().double() - truth).abs().max().item() / s)
        e_ac.append((gac[name].grad.double() - truth).abs().max().item() / s)
    e_gpu, e_ac = np.array(e_gpu), np.array(e_ac)
    assert np.isfinite(e_gpu).all()
    assert np.median(e_gpu) <= 2 * np.median(e_ac) + 1e-3, (np.median(e_gpu), np.median(e_ac))
    assert np.mean(e_gpu) <= 2 * np.mean(e_ac) + 1e-3, (np.mean(e_gpu), np.mean(e_ac))
    assert e_gpu.max() <= 2 * e_ac.max() + 1e-2, (e_gpu.max(), e_ac.max())
    # running statistics were updated from the (rounded) batch statistics: as close to the fp64 buffers as the autocast oracle's are
    b64, bac = dict(ref64.named_buffers()), dict(ref_ac.named_buffers())
    eb_gpu, eb_ac = [], []
    for name, b in net.named_buffers():
        if b.dtype == torch.long:
            assert int(b) == int(b64[name]) == 1, name
            continue
        s = b64[name].abs().max().item() + 1e-30
        eb_gpu.append((b.cpu().double() - b64[name]).abs().max().item() / s)
        eb_ac.append((bac[name].double() - b64[name]).abs().max().item() / s)
    assert np.median(eb_gpu) <= 2 * np.median(eb_ac) + 1e-3 and max(eb_gpu) <= 2 * max(eb_ac) + 1e-2, (np.median(eb_gpu), np.median(eb_ac), max(eb_gpu), max(eb_ac))


@pytest.mark.parametrize("B,H,W", [(3, 96, 160), (1, 224, 96), (5, 64, 64), (7, 128, 32)])
def test_amp_step_at_odd_batches_and_non_square_inputs(B, H, W):
    """The mixed-precision forward + backward at geometries the dispatch rules were not tuned on: forward as close to an fp64 run of the
    oracle as the oracle under torch.autocast is, finite gradients that point where the fp64 gradients point (cosine over the whole
    gradient vector; with maps of a few pixels per channel single ReLU-mask flips move individual tensors by percents)."""
    ref, net = _pair(seed=B + H)
    g = torch.Generator().manual_seed(W + B)
    x = torch.randn(B, 3, H, W, generator=g)
    dy = torch.randn(B, 7, H // 4, W // 4, generator=g) * 0.1
    ref.train(); net.train()
    ref64 = copy.deepcopy(ref).double()
    ref_ac = copy.deepcopy(ref)
    out64 = ref64(x.double()); out64.backward(dy.double())
    with torch.autocast(device_type="cpu", dtype=torch.bfloat16):
        out_ac = ref_ac(x)
    out_ac.float().backward(dy)
    out, tape = net.forward_train(x.to(DEV), amp=True)
    net.backward_from(tape, dy.to(DEV))
    scale = out64.abs().max().item()
    e_gpu = (out.cpu().double() - out64.detach()).abs().max().item() / scale
    e_ac = (out_ac.detach().double() - out64.detach()).abs().max().item() / scale
    assert e_gpu <= 2 * e_ac + 1e-3, (e_gpu, e_ac)
    g64, gac = dict(ref64.named_parameters()), dict(ref_ac.named_parameters())
    gpu = torch.cat([net.grad_of(p).cpu().double().flatten() for _, p in net.named_parameters()])
    tru = torch.cat([g64[n].grad.flatten() for n, _ in net.named_parameters()])
    aut = torch.cat([gac[n].grad.double().flatten() for n, _ in net.named_parameters()])
    assert torch.isfinite(gpu).all()
    cos = lambda a, b: float(torch.dot(a, b) / (a.norm() * b.norm()))
    # the yardstick is the autocast oracle's own distance from the fp64 gradient (bf16 rounding flips many masks on maps this small)
    assert 1 - cos(gpu, tru) <= 2 * (1 - cos(aut, tru)) + 1e-2, (cos(gpu, tru), cos(aut, tru))
    assert 0.8 < float(gpu.norm() / tru.norm()) < 1.25, float(gpu.norm() / tru.norm())


def test_amp_training_reduces_loss_and_cli(tmp_path, monkeypatch, capsys):
    from structuredetector_amd.cli import train
    from structuredetector_amd.data import Encode
    from structuredetector_amd.data.synthetic import synthetic_batch
    from structuredetector_amd.model import Network
    from structuredetector_amd.model.trainer import TrainStep
    from tests.test_host_cpu import make_args
    dev = torch.device(DEV)
    args = make_args(2, 1, 20, 40, device=dev, learning_rate=1e-3, use_amp=True)
    torch.manual_seed(0)
    net = Network(args, pretrained=False).to(dev).train()
    step = TrainStep(net, args)
    assert step.amp
    enc = Encode(args)
    tgt = enc.render(enc.plan(128, 128, *synthetic_batch(np.random.default_rng(0), 8, 128, 128, 2, 1)), dev)
    x = torch.randn(8, 3, 128, 128, device=dev)
    losses = [float(step(x, tgt)[0]) for _ in range(12)]
    assert np.isfinite(losses).all() and losses[-1] < 0.7 * losses[0], losses
    assert net.flat_params.dtype == torch.float32 and net.flat_params_bf16.dtype == torch.bfloat16      # fp32 master weights
    # deterministic: same state -> same step, bit for bit
    a = net.flat_params.clone(); m1 = step.exp_avg.clone(); m2 = step.exp_avg_sq.clone(); sc = step.step_count
    l1 = step(x, tgt).clone(); p1 = net.flat_params.clone()
    net.flat_params.copy_(a); step.exp_avg.copy_(m1); step.exp_avg_sq.copy_(m2); step.step_count = sc
    l2 = step(x, tgt).clone()
    assert torch.equal(l1, l2) and torch.equal(p1, net.flat_params)
    # `train --amp` end to end (validation runs the bf16 inference forward, as the reference's autocast validation does)
    monkeypatch.chdir(tmp_path)
    (tmp_path / "labels.json").write_text(json.dumps({"labels": ["bean", "maize"], "parts": ["leaf"]}))
    train.main(["-W", "128", "-H", "128", "-s", "stem", "--labels", str(tmp_path / "labels.json"), "--synthetic", "16", "-b", "8", "-e", "2", "--amp"])
    out = capsys.readouterr().out
    assert "epoch 1: total" in out and "validation (" in out


@pytest.mark.parametrize("case", [(4, 32, 32, 128, 128, 3, 1, 1), (2, 16, 16, 256, 256, 3, 1, 1), (2, 64, 64, 128, 256, 3, 1, 1), (6, 16, 16, 128, 128, 3, 1, 1),
                                  # maps of 128 pixels and wider: 64-pixel column strips of four rows (2 and 4 strips; 3 row blocks)
                                  (1, 8, 128, 128, 128, 3, 1, 1), (1, 4, 256, 128, 128, 3, 1, 1), (2, 12, 128, 128, 256, 3, 1, 1)])
def test_conv_bf16_two_group_kernel(case):
    """k_conv3x3_bf16_pp (512-pixel x 128-channel tiles, two wave groups half a tap apart, 7-stage weight ring): forced onto small
    grids with sd_set_option, then the same checks as the other bf16 conv kernels (forward + fused statistics, data-gradient with
    the three residual modes) plus the forward epilogue (folded BatchNorm, ReLU, residual, half-size residual upsampled)."""
    from structuredetector_amd import _lib as L
    lib = L.lib()
    B, H, W, cin, cout, k, stride, pad = case
    d = make_desc(L, B, H, W, cin, cout, k, stride, pad)
    L.check(lib.sd_set_option(b"conv_pp_min_tiles", 1))
    L.check(lib.sd_set_option(b"conv_fwd_split_k", 0))
    try:
        assert lib.sd_conv2d_kernel_name(C.byref(d), 16).decode() == "k_conv3x3_bf16_pp"
        assert lib.sd_conv2d_kernel_name(C.byref(d), 17).decode() == "k_conv3x3_bf16_pp"
        test_conv_bf16_forward_statistics_and_data_gradient(case)
        g = torch.Generator().manual_seed(sum(case) + 1)
        x = torch.randn(B, cin, H, W, generator=g).bfloat16().float()
        w = (torch.randn(cout, cin, k, k, generator=g) / (cin * k * k) ** 0.5).bfloat16().float()
        scale = torch.rand(cout, generator=g) + 0.5
        shift = torch.randn(cout, generator=g)
        res = torch.randn(B, cout, H, W, generator=g).bfloat16().float()
        half = torch.randn(B, cout, H // 2, W // 2, generator=g).bfloat16().float()
        xd, wd = nhwc16(x), nhwc16(w)
        scale_d, shift_d = scale.to(DEV), shift.to(DEV)                       # (named: a temporary would be freed before the launch)
        y = torch.empty(B, H, W, cout, dtype=torch.bfloat16, device=DEV)
        conv = F.conv2d(x, w, None, stride, pad) * scale.view(1, -1, 1, 1) + shift.view(1, -1, 1, 1)
        for r16, up2, relu, ref in ((nhwc16(res), 0, 1, torch.relu(conv + res)),
                                    (nhwc16(half), 1, 0, conv + F.interpolate(half, scale_factor=2, mode="nearest")),
                                    (None, 0, 0, conv)):
            L.check(lib.sd_conv2d_fwd_bf16(xd.data_ptr(), wd.data_ptr(), y.data_ptr(), C.byref(d), scale_d.data_ptr(), shift_d.data_ptr(),
                                           r16.data_ptr() if r16 is not None else 0, up2, relu, 0, 0, L.stream()))
            close(back(y), ref, 8e-3)
        # against the one-group kernel on the same operands: the two-group kernel multiplies on v_mfma_f32_16x16x32_bf16, the one-group
        # kernel on 32x32x16 -- the same fp32 products summed in another order inside the instruction: equal up to the last bf16 bit
        L.check(lib.sd_set_option(b"conv_pp_min_tiles", 1 << 30))
        L.check(lib.sd_set_option(b"conv_patch_min_tiles", 1))
        y1 = torch.empty_like(y)
        L.check(lib.sd_conv2d_fwd_bf16(xd.data_ptr(), wd.data_ptr(), y1.data_ptr(), C.byref(d), scale_d.data_ptr(), shift_d.data_ptr(),
                                       0, 0, 0, 0, 0, L.stream()))
        if lib.sd_conv2d_kernel_name(C.byref(d), 16).decode().startswith("k_conv3x3_patch"):
            ya, yb = y.float(), y1.float()
            assert (ya - yb).abs().max() <= 2.0 ** -7 * yb.abs().max()                 # one bf16 ulp of the largest value
            assert (ya != yb).float().mean() < 0.05                                  # and only where a sum sits on a rounding boundary
    finally:
        L.check(lib.sd_set_option(b"conv_pp_min_tiles", 200))
        L.check(lib.sd_set_option(b"conv_patch_min_tiles", 512))
        L.check(lib.sd_set_option(b"conv_fwd_split_k", 1))


@pytest.mark.parametrize("case", [(2, 16, 16, 64), (1, 8, 20, 64), (3, 12, 8, 128), (2, 32, 32, 128), (64, 128, 128, 64)])
def test_conv1x1_stream_kernel(case):
    """k_conv1x1_stream_bf16 (the FPN laterals: 1x1 / stride 1 onto 128 channels, weights in registers, pixels streamed from global memory
    as MFMA operands, 16-byte epilogue without LDS): bias, residual, half-size residual upsampled (network.py:13-18), ReLU; identical
    maths to the tile kernel (bf16 products, fp32 sums) -> equal to one bf16 ulp; the 1x1 data-gradient reaches it too."""
    from structuredetector_amd import _lib as L
    lib = L.lib()
    B, H, W, cin = case
    cout = 128
    d = make_desc(L, B, H, W, cin, cout, 1, 1, 0)
    g = torch.Generator().manual_seed(sum(case))
    x = torch.randn(B, cin, H, W, generator=g).bfloat16().float()
    w = (torch.randn(cout, cin, 1, 1, generator=g) / cin ** 0.5).bfloat16().float()
    bias = torch.randn(cout, generator=g)
    res = torch.randn(B, cout, H, W, generator=g).bfloat16().float()
    half = torch.randn(B, cout, H // 2, W // 2, generator=g).bfloat16().float()
    xd, wd, bias_d = nhwc16(x), nhwc16(w), bias.to(DEV)
    conv = F.conv2d(x, w, bias)
    L.check(lib.sd_set_option(b"conv1x1_stream_min_pixels", 32))
    try:
        assert lib.sd_conv2d_kernel_name(C.byref(d), 16).decode().startswith("k_conv1x1_stream_bf16")
        outs = []
        for r16, up2, relu, ref in ((None, 0, 0, conv), (nhwc16(res), 0, 1, torch.relu(conv + res)),
                                    (nhwc16(half), 1, 0, conv + F.interpolate(half, scale_factor=2, mode="nearest"))):
            y = torch.full((B, H, W, cout), float("nan"), dtype=torch.bfloat16, device=DEV)
            L.check(lib.sd_conv2d_fwd_bf16(xd.data_ptr(), wd.data_ptr(), y.data_ptr(), C.byref(d), 0, bias_d.data_ptr(),
                                           r16.data_ptr() if r16 is not None else 0, up2, relu, 0, 0, L.stream()))
            close(back(y), ref, 8e-3)
            outs.append(y)
        # the tile kernel on the same operands
        L.check(lib.sd_set_option(b"conv1x1_stream_min_pixels", 1 << 30))
        assert lib.sd_conv2d_kernel_name(C.byref(d), 16).decode().startswith("k_conv_igemm")
        y1 = torch.empty_like(outs[0])
        L.check(lib.sd_conv2d_fwd_bf16(xd.data_ptr(), wd.data_ptr(), y1.data_ptr(), C.byref(d), 0, bias_d.data_ptr(), 0, 0, 0, 0, 0, L.stream()))
        ya, yb = outs[0].float(), y1.float()
        assert (ya - yb).abs().max() <= 2.0 ** -7 * yb.abs().max() and (ya != yb).float().mean() < 0.05
        # data gradient of a 128 -> 128 1x1 conv = the same stream with the transposed weights (+ an accumulated gradient)
        if cin == 128:
            L.check(lib.sd_set_option(b"conv1x1_stream_min_pixels", 32))
            assert lib.sd_conv2d_kernel_name(C.byref(d), 17).decode().startswith("k_conv1x1_stream_bf16")
            dy = torch.randn(B, cout, H, W, generator=g).bfloat16().float()
            wt = w.permute(1, 2, 3, 0).contiguous().to(DEV).to(torch.bfloat16)
            dx = torch.empty(B, H, W, cin, dtype=torch.bfloat16, device=DEV)
            prev = nhwc16(res)
            L.check(lib.sd_conv2d_dgrad_bf16(nhwc16(dy).data_ptr(), wt.data_ptr(), dx.data_ptr(), C.byref(d), prev.data_ptr(), 1, L.stream()))
            xg = x.clone().requires_grad_(True)
            F.conv2d(xg, w).backward(dy)
            close(back(dx), xg.grad + res, 8e-3)
    finally:
        L.check(lib.sd_set_option(b"conv1x1_stream_min_pixels", 32 * 2048))


def test_conv_bf16_narrow_patch_tiles_for_wide_layers():
    """bf16 layers whose 128-channel patch tiles do not fill the chip (layer4 at bs=64) take 64-channel patch tiles (`conv_patch_narrow` = 2):
    the checks of the other bf16 conv kernels on a 256 -> 256 channel layer cut into four channel tiles."""
    from structuredetector_amd import _lib as L
    lib = L.lib()
    case = (4, 16, 16, 256, 256, 3, 1, 1)
    d = make_desc(L, *case)
    L.check(lib.sd_set_option(b"conv_patch_min_tiles", 12))          # 4 x 2 tiles of 128 channels < 12 <= 4 x 4 tiles of 64
    L.check(lib.sd_set_option(b"conv_fwd_split_k", 0))
    try:
        assert lib.sd_conv2d_kernel_name(C.byref(d), 16).decode() == "k_conv3x3_patch<64, true>"
        assert lib.sd_conv2d_kernel_name(C.byref(d), 17).decode() == "k_conv3x3_patch<64, true>"
        L.check(lib.sd_set_option(b"conv_patch_narrow", 1))
        assert lib.sd_conv2d_kernel_name(C.byref(d), 16).decode() == "k_conv_igemm<128, 0, true>"
        L.check(lib.sd_set_option(b"conv_patch_narrow", 2))
        test_conv_bf16_forward_statistics_and_data_gradient(case)
    finally:
        L.check(lib.sd_set_option(b"conv_patch_min_tiles", 512))
        L.check(lib.sd_set_option(b"conv_patch_narrow", 2))
        L.check(lib.sd_set_option(b"conv_fwd_split_k", 1))


@pytest.mark.parametrize("case", [(2, 6, 10, 128, 7), (3, 32, 40, 128, 8), (2, 16, 16, 64, 5), (1, 128, 128, 128, 7)])
def test_head_backward_on_bf16_activations(case):
    """sd_head_bwd_bf16 (bf16 FPN output in, bf16 input gradient out, fp32 weight / bias gradients) against PyTorch on the same bf16-rounded
    activation: dx within one bf16 rounding, dw / dbias to fp32 accuracy.  Pixel counts that are not multiples of 64 / 1024 included."""
    from structuredetector_amd import _lib as L
    lib = L.lib()
    B, H, W, Cc, Co = case
    g = torch.Generator().manual_seed(sum(case))
    x = torch.randn(B, Cc, H, W, generator=g).bfloat16().float().requires_grad_(True)
    w = (torch.randn(Co, Cc, 1, 1, generator=g) / 11).requires_grad_(True)
    b = torch.randn(Co, generator=g).requires_grad_(True)
    dy = torch.randn(B, Co, H, W, generator=g)
    F.conv2d(x, w, b).backward(dy)
    x16 = nhwc16(x.detach())
    dy_d, w_d = dy.to(DEV), w.detach().reshape(Co, Cc).to(DEV)
    dx = torch.empty(B, H, W, Cc, dtype=torch.bfloat16, device=DEV)
    dw = torch.empty(Co, Cc, device=DEV); db = torch.empty(Co, device=DEV)
    ws = torch.empty(max(lib.sd_head_bwd_workspace_bytes(B, H * W, Cc, Co), 256), dtype=torch.uint8, device=DEV)
    L.check(lib.sd_head_bwd_bf16(dy_d.data_ptr(), x16.data_ptr(), w_d.data_ptr(), dx.data_ptr(), dw.data_ptr(), db.data_ptr(), B, H * W, Cc, Co, 0,
                                 ws.data_ptr(), ws.numel(), L.stream()))
    close(back(dx), x.grad, 5e-3)
    close(dw.cpu(), w.grad.reshape(Co, Cc), 2e-5)
    close(db.cpu(), b.grad, 2e-5)
    # accumulate = 1 adds onto the existing gradients
    L.check(lib.sd_head_bwd_bf16(dy_d.data_ptr(), x16.data_ptr(), w_d.data_ptr(), dx.data_ptr(), dw.data_ptr(), db.data_ptr(), B, H * W, Cc, Co, 1,
                                 ws.data_ptr(), ws.numel(), L.stream()))
    close(dw.cpu(), 2 * w.grad.reshape(Co, Cc), 2e-5)


@pytest.mark.parametrize("rows16", [1, 0])
@pytest.mark.parametrize("case", [(2, 16, 128, 64, 64, 3, 1, 1), (1, 24, 256, 64, 64, 3, 1, 1), (3, 9, 128, 64, 64, 3, 1, 1), (8, 40, 128, 64, 64, 3, 1, 1),
                                  (2, 1, 128, 64, 64, 3, 1, 1), (1, 2, 128, 64, 64, 3, 1, 1)])
def test_conv_bf16_row_stream_kernel_64_channels(case, rows16):
    """layer1's 64 -> 64 convs as a row stream (persistent blocks walk the rows of a 128-pixel strip, weights in registers): forced onto small
    problems with sd_set_option, then the checks of the other bf16 conv kernels (forward + fused statistics, data-gradient plain / + residual)
    plus the forward epilogue (affine, + residual, ReLU).  rows16 = 1: k_conv3x3_c64_rows16_bf16 (round 5: swapped MFMA operands on
    16x16x32, epilogue of the previous row in registers out of a second accumulator set, one instantiation per epilogue kind);
    rows16 = 0: k_conv3x3_c64_rows_bf16 (32x32x16, output stored one row late through an LDS scratch).
    Shapes: two strips per row (W = 256), row counts that are not a multiple of the unit, several units per strip, one- and two-row maps
    (a unit whose first row is its last / whose second accumulator set is the last one read)."""
    from structuredetector_amd import _lib as L
    lib = L.lib()
    B, H, W, cin, cout, k, stride, pad = case
    d = make_desc(L, B, H, W, cin, cout, k, stride, pad)
    L.check(lib.sd_set_option(b"conv_rows64_min_units", 1))
    L.check(lib.sd_set_option(b"conv_fwd_split_k", 0))
    L.check(lib.sd_set_option(b"conv_rows16", rows16))
    name = "k_conv3x3_c64_rows16_bf16" if rows16 else "k_conv3x3_c64_rows_bf16"
    try:
        assert lib.sd_conv2d_kernel_name(C.byref(d), 16).decode() == name
        assert lib.sd_conv2d_kernel_name(C.byref(d), 17).decode() == name
        test_conv_bf16_forward_statistics_and_data_gradient(case)
        g = torch.Generator().manual_seed(sum(case) + 2)
        x = torch.randn(B, cin, H, W, generator=g).bfloat16().float()
        w = (torch.randn(cout, cin, k, k, generator=g) / (cin * k * k) ** 0.5).bfloat16().float()
        scale = torch.rand(cout, generator=g) + 0.5
        shift = torch.randn(cout, generator=g)
        res = torch.randn(B, cout, H, W, generator=g).bfloat16().float()
        xd, wd, res_d = nhwc16(x), nhwc16(w), nhwc16(res)
        scale_d, shift_d = scale.to(DEV), shift.to(DEV)
        y = torch.empty(B, H, W, cout, dtype=torch.bfloat16, device=DEV)
        conv = F.conv2d(x, w, None, stride, pad) * scale.view(1, -1, 1, 1) + shift.view(1, -1, 1, 1)
        for r16, relu, ref in ((res_d, 1, torch.relu(conv + res)), (None, 1, torch.relu(conv)), (None, 0, conv)):
            L.check(lib.sd_conv2d_fwd_bf16(xd.data_ptr(), wd.data_ptr(), y.data_ptr(), C.byref(d), scale_d.data_ptr(), shift_d.data_ptr(),
                                           r16.data_ptr() if r16 is not None else 0, 0, relu, 0, 0, L.stream()))
            close(back(y), ref, 8e-3)
    finally:
        L.check(lib.sd_set_option(b"conv_rows64_min_units", 192))
        L.check(lib.sd_set_option(b"conv_fwd_split_k", 1))
        L.check(lib.sd_set_option(b"conv_rows16", 1))


def test_stride2_convs_on_the_256_row_bf16_tiles():
    """k_conv_igemm_big<128, 0 / 2, true> (opt-in, sd_set_option("igemm_big_bf16", 1): the 3x3 / stride 2 convs of the mixed-precision step on 512+
    tiles of 256 rows, three LDS stages, counted waits) against k_conv_igemm<.., true> on the same operands (bf16 products, fp32 sums, another summation order) and,
    on one image, against torch: forward with bias + ReLU and with fused BatchNorm statistics, data-gradient with a residual."""
    from structuredetector_amd import _lib as L
    lib = L.lib()
    B, H, W, cin, cout = 16, 128, 128, 128, 256
    d = make_desc(L, B, H, W, cin, cout, 3, 2, 1)
    g = torch.Generator().manual_seed(77)
    x = (torch.randn(B, cin, H, W, generator=g)).bfloat16()
    w = (torch.randn(cout, cin, 3, 3, generator=g) / (9 * cin) ** 0.5).bfloat16()
    bias = torch.randn(cout, generator=g)
    xd = x.permute(0, 2, 3, 1).contiguous().to(DEV); wd = w.permute(0, 2, 3, 1).contiguous().to(DEV); bias_d = bias.to(DEV)
    dy = torch.randn(B, cout, H // 2, W // 2, generator=g).bfloat16()
    dyd = dy.permute(0, 2, 3, 1).contiguous().to(DEV)
    wt = w.permute(1, 2, 3, 0).contiguous().to(DEV)
    prev = torch.randn(B, H, W, cin, generator=g).bfloat16().to(DEV)
    ws = torch.empty(max(lib.sd_conv2d_fwd_bf16_bn_stats_workspace_bytes(C.byref(d)), 256), dtype=torch.uint8, device=DEV)
    out = {}
    try:
        for big in (1, 0):
            L.check(lib.sd_set_option(b"igemm_big_bf16", big))
            names = [lib.sd_conv2d_kernel_name(C.byref(d), p).decode() for p in (16, 17)]
            assert names == (["k_conv_igemm_big<128, 0, true>", "k_conv_igemm_big<128, 2, true>"] if big else
                             ["k_conv_igemm<128, 0, true>", "k_conv_igemm<128, 2, true>"]), names
            y = torch.full((B, H // 2, W // 2, cout), float("nan"), dtype=torch.bfloat16, device=DEV)
            L.check(lib.sd_conv2d_fwd_bf16(xd.data_ptr(), wd.data_ptr(), y.data_ptr(), C.byref(d), 0, bias_d.data_ptr(), 0, 0, 1, 0, 0, L.stream()))
            ys = torch.full_like(y, float("nan"))
            mean = torch.empty(cout, device=DEV); invstd = torch.empty(cout, device=DEV)
            ws.fill_(0xFF)                 # (NaNs: a partial row the kernel does not write must not be read)
            L.check(lib.sd_conv2d_fwd_bf16_bn_stats(xd.data_ptr(), wd.data_ptr(), ys.data_ptr(), C.byref(d), 1e-5, 0.1, 0, 0, mean.data_ptr(), invstd.data_ptr(),
                                                    ws.data_ptr(), ws.numel(), L.stream()))
            dx = torch.full((B, H, W, cin), float("nan"), dtype=torch.bfloat16, device=DEV)
            L.check(lib.sd_conv2d_dgrad_bf16(dyd.data_ptr(), wt.data_ptr(), dx.data_ptr(), C.byref(d), prev.data_ptr(), 1, L.stream()))
            out[big] = (y.float(), ys.float(), mean.clone(), invstd.clone(), dx.float())
    finally:
        L.check(lib.sd_set_option(b"igemm_big_bf16", 0))
    for a, b in zip(out[1], out[0]):
        assert torch.isfinite(a).all()
        assert (a - b).abs().max() <= 2.0 ** -7 * b.abs().max() + 1e-6, float((a - b).abs().max())
    # one image against torch (fp32 on the bf16-representable operands)
    ref = torch.relu(F.conv2d(x[:1].float(), w.float(), bias, 2, 1))
    close(out[1][0][:1].permute(0, 3, 1, 2).cpu(), ref, 8e-3)
    xg = x[:1].float().requires_grad_(True)
    F.conv2d(xg, w.float(), None, 2, 1).backward(dy[:1].float())
    close(out[1][4][:1].permute(0, 3, 1, 2).cpu(), xg.grad + prev[:1].float().permute(0, 3, 1, 2).cpu(), 8e-3)


@pytest.mark.parametrize("shape", [(2, 16, 24), (3, 64, 32), (1, 256, 256), (2, 36, 44), (1, 18, 22)])
def test_stem_tail_on_bf16_activations_equals_fp32_kernels(shape):
    """sd_bn_relu_maxpool_fwd_bf16 / sd_maxpool_bn_relu_bwd_bf16 (mixed-precision stem tail: bf16 conv output, pooled map and pooled
    gradient; fp32 arithmetic) against the fp32 kernels on the widened tensors: same winning taps, pooled map = the fp32 result rounded
    once, input gradient / dgamma / dbeta bit-identical."""
    from structuredetector_amd import _lib as L
    lib = L.lib()
    B, H, W = shape
    Cc = 64
    g = torch.Generator().manual_seed(B + H + W)
    x16 = torch.randn(B, H, W, Cc, generator=g).bfloat16().to(DEV)
    x32 = x16.float()
    mean = torch.randn(Cc, generator=g).to(DEV) * 0.1
    invstd = (torch.rand(Cc, generator=g) + 0.5).to(DEV)
    gamma = (torch.rand(Cc, generator=g) + 0.5).to(DEV)
    beta = (torch.randn(Cc, generator=g) * 0.2).to(DEV)
    Hp, Wp = (H + 2 - 3) // 2 + 1, (W + 2 - 3) // 2 + 1
    y32 = torch.empty(B, Hp, Wp, Cc, device=DEV); i32 = torch.empty(B, Hp, Wp, Cc, dtype=torch.uint8, device=DEV)
    y16 = torch.empty(B, Hp, Wp, Cc, dtype=torch.bfloat16, device=DEV); i16 = torch.empty_like(i32)
    L.check(lib.sd_bn_relu_maxpool_fwd(x32.data_ptr(), B, H, W, Cc, mean.data_ptr(), invstd.data_ptr(), gamma.data_ptr(), beta.data_ptr(),
                                       y32.data_ptr(), i32.data_ptr(), L.stream()))
    L.check(lib.sd_bn_relu_maxpool_fwd_bf16(x16.data_ptr(), B, H, W, Cc, mean.data_ptr(), invstd.data_ptr(), gamma.data_ptr(), beta.data_ptr(),
                                            y16.data_ptr(), i16.data_ptr(), L.stream()))
    assert torch.equal(i16, i32)
    assert torch.equal(y16, y32.bfloat16())
    # one window per thread (sd_set_option("pool_fwd_pair", 0)) = two windows per thread, bit for bit
    L.check(lib.sd_set_option(b"pool_fwd_pair", 0))
    try:
        y32b = torch.empty_like(y32); i32b = torch.empty_like(i32); y16b = torch.empty_like(y16); i16b = torch.empty_like(i16)
        L.check(lib.sd_bn_relu_maxpool_fwd(x32.data_ptr(), B, H, W, Cc, mean.data_ptr(), invstd.data_ptr(), gamma.data_ptr(), beta.data_ptr(),
                                           y32b.data_ptr(), i32b.data_ptr(), L.stream()))
        L.check(lib.sd_bn_relu_maxpool_fwd_bf16(x16.data_ptr(), B, H, W, Cc, mean.data_ptr(), invstd.data_ptr(), gamma.data_ptr(), beta.data_ptr(),
                                                y16b.data_ptr(), i16b.data_ptr(), L.stream()))
    finally:
        L.check(lib.sd_set_option(b"pool_fwd_pair", 1))
    assert torch.equal(y32b, y32) and torch.equal(i32b, i32) and torch.equal(y16b, y16) and torch.equal(i16b, i16)
    dp16 = torch.randn(B, Hp, Wp, Cc, generator=g).bfloat16().to(DEV)
    dp32 = dp16.float()
    ws = torch.empty(max(lib.sd_col_reduce_workspace_bytes(B * H * W, Cc), 256), dtype=torch.uint8, device=DEV)
    out = {}
    for tag, fn, dp, xx in (("f32", lib.sd_maxpool_bn_relu_bwd, dp32, x32), ("bf16", lib.sd_maxpool_bn_relu_bwd_bf16, dp16, x16)):
        dx = torch.empty(B, H, W, Cc, device=DEV); dg = torch.empty(Cc, device=DEV); db = torch.empty(Cc, device=DEV)
        L.check(fn(dp.data_ptr(), i32.data_ptr(), xx.data_ptr(), B, H, W, Cc, mean.data_ptr(), invstd.data_ptr(), gamma.data_ptr(), beta.data_ptr(),
                   dx.data_ptr(), dg.data_ptr(), db.data_ptr(), 0, ws.data_ptr(), ws.numel(), L.stream()))
        out[tag] = (dx, dg, db)
    for a, b in zip(out["f32"], out["bf16"]):
        assert torch.equal(a, b)
    # bf16 input gradient (what the step uses): the same values rounded once, the same dgamma / dbeta
    dx16 = torch.empty(B, H, W, Cc, dtype=torch.bfloat16, device=DEV); dg = torch.empty(Cc, device=DEV); db = torch.empty(Cc, device=DEV)
    L.check(lib.sd_maxpool_bn_relu_bwd_bf16_dx16(dp16.data_ptr(), i32.data_ptr(), x16.data_ptr(), B, H, W, Cc, mean.data_ptr(), invstd.data_ptr(),
                                                 gamma.data_ptr(), beta.data_ptr(), dx16.data_ptr(), dg.data_ptr(), db.data_ptr(), 0, ws.data_ptr(),
                                                 ws.numel(), L.stream()))
    assert torch.equal(dx16, out["f32"][0].bfloat16()) and torch.equal(dg, out["f32"][1]) and torch.equal(db, out["f32"][2])


@pytest.mark.parametrize("shape", [(2, 64, 96), (1, 512, 512), (3, 160, 288), (5, 32, 32)])
def test_stem_weight_gradient_on_the_bf16_mfma(shape):
    """sd_conv2d_stem_wgrad_bf16mm (pixel-major operands transposed on their way into LDS, even / odd column planes of the image patch in
    four shifted copies) against the fp32 kernel on bf16-rounded operands (products of two bf16 are exact in fp32: only the summation order
    differs) and against autograd.  Shapes: partial last 128-pixel tile, two tiles per row, a map smaller than a tile."""
    from structuredetector_amd import _lib as L
    lib = L.lib()
    B, H, W = shape
    g = torch.Generator().manual_seed(H * 3 + W)
    img = torch.randn(B, 3, H, W, generator=g).bfloat16().float()
    d0 = make_desc(L, B, H, W, 3, 64, 7, 2, 3)
    dy = torch.randn(B, 64, d0.Ho, d0.Wo, generator=g).bfloat16().float()
    img_d = img.to(DEV)
    dy_d = dy.permute(0, 2, 3, 1).contiguous().to(DEV)
    ws = torch.empty(max(lib.sd_conv2d_stem_wgrad_workspace_bytes(C.byref(d0)), 256), dtype=torch.uint8, device=DEV)
    dw32 = torch.empty(64, 7, 7, 3, device=DEV); dw16 = torch.empty_like(dw32)
    L.check(lib.sd_conv2d_stem_wgrad(dy_d.data_ptr(), img_d.data_ptr(), dw32.data_ptr(), C.byref(d0), 0, ws.data_ptr(), ws.numel(), L.stream()))
    L.check(lib.sd_conv2d_stem_wgrad_bf16mm(dy_d.data_ptr(), img_d.data_ptr(), dw16.data_ptr(), C.byref(d0), 0, ws.data_ptr(), ws.numel(), L.stream()))
    close(dw16.cpu(), dw32.cpu(), 2e-5)
    w = torch.zeros(64, 3, 7, 7, requires_grad=True)
    F.conv2d(img, w, None, 2, 3).backward(dy)
    close(dw16.permute(0, 3, 1, 2).cpu(), w.grad, 2e-5)
    # accumulate = 1
    L.check(lib.sd_conv2d_stem_wgrad_bf16mm(dy_d.data_ptr(), img_d.data_ptr(), dw16.data_ptr(), C.byref(d0), 1, ws.data_ptr(), ws.numel(), L.stream()))
    close(dw16.permute(0, 3, 1, 2).cpu(), 2 * w.grad, 2e-5)


@pytest.mark.parametrize("shape", [(2, 64, 96), (1, 512, 512), (3, 160, 288), (5, 32, 32), (2, 136, 520), (1, 1024, 256)])
def test_stem_forward_row_ring(shape):
    """k_stem_fwd_bf16_ring (mixed-precision stem forward: bf16 NHWC output + BatchNorm statistics; image rows by LDS-DMA into a ring of
    shifted bf16 copies, weights in registers, reduction ordered (channel, row, column)) against k_stem_fwd<true> on the same operands
    (another summation order: equal to one bf16 ulp) and against torch on the bf16-rounded operands."""
    from structuredetector_amd import _lib as L
    lib = L.lib()
    B, H, W = shape
    g = torch.Generator().manual_seed(H * 7 + W)
    img = torch.randn(B, 3, H, W, generator=g)
    w = torch.randn(64, 3, 7, 7, generator=g) / 12
    d0 = make_desc(L, B, H, W, 3, 64, 7, 2, 3)
    img_d = img.to(DEV); w_d = w.permute(0, 2, 3, 1).contiguous().to(DEV)
    ws = torch.empty(max(lib.sd_conv2d_stem_fwd_bn_stats_workspace_bytes(C.byref(d0)), 256), dtype=torch.uint8, device=DEV)
    out = {}
    try:
        for ring in (1, 0):
            L.check(lib.sd_set_option(b"stem_fwd_ring", ring))
            y = torch.full((B, d0.Ho, d0.Wo, 64), float("nan"), dtype=torch.bfloat16, device=DEV)
            mean = torch.empty(64, device=DEV); invstd = torch.empty(64, device=DEV)
            rm = torch.zeros(64, device=DEV); rv = torch.ones(64, device=DEV)
            ws.fill_(0xFF)
            L.check(lib.sd_conv2d_stem_fwd_bn_stats_bf16(img_d.data_ptr(), w_d.data_ptr(), y.data_ptr(), C.byref(d0), 1e-5, 0.1, rm.data_ptr(), rv.data_ptr(),
                                                         mean.data_ptr(), invstd.data_ptr(), ws.data_ptr(), ws.numel(), L.stream()))
            out[ring] = (y.float(), mean, invstd, rm, rv)
    finally:
        L.check(lib.sd_set_option(b"stem_fwd_ring", 1))
    ya, yb = out[1][0], out[0][0]
    assert torch.isfinite(ya).all()
    assert (ya - yb).abs().max() <= 2.0 ** -7 * yb.abs().max() and (ya != yb).float().mean() < 0.05
    for a, b in zip(out[1][1:], out[0][1:]):
        np.testing.assert_allclose(a.cpu().numpy(), b.cpu().numpy(), rtol=2e-4, atol=2e-5)
    ref = F.conv2d(img.bfloat16().float(), w.bfloat16().float(), None, 2, 3)
    close(ya.permute(0, 3, 1, 2).cpu(), ref, 8e-3)


@pytest.mark.parametrize("shape", [(2, 64, 96), (1, 512, 512), (3, 160, 288), (5, 32, 32), (2, 136, 520), (1, 1024, 256)])
def test_stem_weight_gradient_from_a_bf16_gradient_row_ring(shape):
    """sd_conv2d_stem_wgrad_bf16 (bf16 dy by LDS-DMA + transposed reads, a block walks down a 128-pixel column strip and keeps a ring of 8
    image rows per channel in its plane copies) against the fp32 kernel on the same bf16-representable operands and against autograd.
    Shapes: partial last 128-pixel tile, several row groups of 32 with a partial last one (Ho = 68, 80, 512), a map smaller than a tile."""
    from structuredetector_amd import _lib as L
    lib = L.lib()
    B, H, W = shape
    g = torch.Generator().manual_seed(H * 5 + W)
    img = torch.randn(B, 3, H, W, generator=g).bfloat16().float()
    d0 = make_desc(L, B, H, W, 3, 64, 7, 2, 3)
    dy = torch.randn(B, 64, d0.Ho, d0.Wo, generator=g).bfloat16().float()
    img_d = img.to(DEV)
    dy32 = dy.permute(0, 2, 3, 1).contiguous().to(DEV)
    dy16 = dy32.bfloat16()
    ws = torch.empty(max(lib.sd_conv2d_stem_wgrad_workspace_bytes(C.byref(d0)), 256), dtype=torch.uint8, device=DEV)
    dw32 = torch.empty(64, 7, 7, 3, device=DEV); dw16 = torch.empty_like(dw32)
    L.check(lib.sd_conv2d_stem_wgrad(dy32.data_ptr(), img_d.data_ptr(), dw32.data_ptr(), C.byref(d0), 0, ws.data_ptr(), ws.numel(), L.stream()))
    L.check(lib.sd_conv2d_stem_wgrad_bf16(dy16.data_ptr(), img_d.data_ptr(), dw16.data_ptr(), C.byref(d0), 0, ws.data_ptr(), ws.numel(), L.stream()))
    close(dw16.cpu(), dw32.cpu(), 2e-5)
    w = torch.zeros(64, 3, 7, 7, requires_grad=True)
    F.conv2d(img, w, None, 2, 3).backward(dy)
    close(dw16.permute(0, 3, 1, 2).cpu(), w.grad, 2e-5)
    L.check(lib.sd_conv2d_stem_wgrad_bf16(dy16.data_ptr(), img_d.data_ptr(), dw16.data_ptr(), C.byref(d0), 1, ws.data_ptr(), ws.numel(), L.stream()))
    close(dw16.permute(0, 3, 1, 2).cpu(), 2 * w.grad, 2e-5)


def test_rows16_kernel_is_bit_identical_to_the_32x32x16_row_stream_on_random_cases():
    """tools/rows16_fuzz.py: 80 random (batch, height incl. 1 / 2 / odd / not a multiple of the unit, one or two strips, forward / data-gradient,
    affine / residual / ReLU) cases; k_conv3x3_c64_rows16_bf16 must reproduce k_conv3x3_c64_rows_bf16 bit for bit (the same fp32 sums: K = 32
    per MFMA instead of 16 does not change the order within a k step here) and itself on a second run -- the check that exposed nothing in
    round 5's final build but would have caught the AGPR-copy hazard of its first (run-to-run different sums from the second launch on)."""
    import subprocess
    import sys
    from pathlib import Path
    root = Path(__file__).resolve().parent.parent
    r = subprocess.run([sys.executable, str(root / "tools" / "rows16_fuzz.py"), "80", "5"], capture_output=True, text=True, timeout=600, cwd=root)
    assert r.returncode == 0 and "80 cases, 0 mismatches" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]
