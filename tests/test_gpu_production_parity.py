"""Whole-network parity AT THE PRODUCTION BATCH (BASELINE configs[2]: bs=64, 512x512, fp32, 2 labels / 1 part; configs[4]
geometry for the bf16 forward): the dispatch rules that only engage at these sizes -- layer1 on the row-stream kernel, the
>= 128-channel layers on patch tiles, layer4 / up2.conv on narrow patch tiles, the all-taps weight gradient -- inside the real
`TrainStep`, against the CPU oracle (src/sdnet/model/network.py:32-84, trainer.py:113-124, loss.py:17-50).

Two layers of evidence:
  * the chain: head output, loss, d(loss)/d(head), BatchNorm batch / running statistics and every parameter gradient of ONE
    production step against the oracle's autograd run of the same step;
  * per kernel family, chaos-free: the oracle's OWN per-layer tensors (input, output, output gradient, input gradient, captured
    by module hooks during that run) are handed to each HIP conv kernel at its production geometry, so forward / data-gradient /
    weight-gradient of every one of the 44 conv layers is compared on identical operands (no ReLU-mask flips, no upstream error),
    bucketed by the device kernel `sd_conv2d_kernel_name` reports, with an fp64 run of one layer per family as the yardstick for
    the 10^6-term weight-gradient sums.
"""
import copy
import ctypes as C
import os
import time
from argparse import Namespace

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import sdnet_oracle as O
from tests.helpers import oracle_conv_trace
from tests.test_gpu_network import close, make_desc

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _rel(got, want):
    """max |got - want| / max |want| (both on the same device)"""
    return float((got - want).abs().max() / (want.abs().max() + 1e-30))


def _names(lib, d):
    return tuple(lib.sd_conv2d_kernel_name(C.byref(d), which).decode() for which in (0, 1, 2))


# the device kernels the production batch must engage ("What's weak" 1): layer -> (forward, data-gradient, weight-gradient)
EXPECTED_KERNELS = {
    "down1.0.conv1": ("k_conv3x3_c64_rows_f32", "k_conv3x3_c64_rows_f32", "k_wgrad3x3_ring2"),
    "down1.2.conv2": ("k_conv3x3_c64_rows_f32", "k_conv3x3_c64_rows_f32", "k_wgrad3x3_ring2"),
    "down2.1.conv1": ("k_conv3x3_patch<128, false>", "k_conv3x3_patch<128, false>", "k_wgrad3x3_ring2"),
    "down3.2.conv2": ("k_conv3x3_patch<128, false>", "k_conv3x3_patch<128, false>", "k_wgrad3x3_ring2"),
    "down4.1.conv1": ("k_conv3x3_patch<64, false>", "k_conv3x3_patch<64, false>", "k_wgrad3x3<16>"),
    "up2.conv.0": ("k_conv3x3_patch<64, false>", "k_conv3x3_patch<64, false>", "k_wgrad3x3_ring2"),
    "up4.conv.0": ("k_conv3x3_patch_roll<128, false>", "k_conv3x3_patch_roll<128, false>", "k_wgrad3x3_ring2"),    # 128-wide map: rolling-buffer entry point
}


def test_train_step_bs64_512_vs_oracle_per_kernel_family():
    from structuredetector_amd import _lib as L
    from structuredetector_amd.data import Encode
    from structuredetector_amd.model import Network
    from structuredetector_amd.model.trainer import TrainStep
    from tests.test_host_cpu import make_args, to_annotation
    lib = L.lib()
    B, S, M, N, K, P = 64, 512, 2, 1, 20, 40
    t0 = time.time()
    torch.set_num_threads(min(16, os.cpu_count() or 1))
    ref = O.build_reference_network(M, N, seed=31).train()
    ref64 = copy.deepcopy(ref).double()
    args = make_args(M, N, K, P, device=torch.device(DEV), learning_rate=1e-3)
    net = Network(args, pretrained=False, raw_output=True)
    net.load_state_dict(ref.state_dict())
    net = net.to(DEV).train()
    g = torch.Generator().manual_seed(64512)
    x = torch.randn(B, 3, S, S, generator=g)
    rng = np.random.default_rng(64)
    scenes = [O.synthetic_scene(rng, S, S, M, N) for _ in range(B)]
    target_o = O.collate([O.encode(S, S, s, M, N, K, P, 4.0, 0.1) for s in scenes])
    target_g = Encode(args).batch((S, S), [to_annotation(args, s, f"i{i}") for i, s in enumerate(scenes)], DEV)

    # ---- the production step (real TrainStep; lr = 0 keeps the weights so that the per-layer part below sees the same ones) ----
    step = TrainStep(net, args, lr=0.0)
    cap = {}
    fwd0, bwd0 = net.forward_train, net.backward_from

    def fwd(images, amp=False):
        cap["head"], cap["tape"] = fwd0(images, amp=amp)
        return cap["head"], cap["tape"]

    def bwd(tape, dhead, on_stage=None):
        cap["dhead"] = dhead.clone()
        return bwd0(tape, dhead, on_stage)

    net.forward_train, net.backward_from = fwd, bwd
    xd = x.to(DEV)
    flat_before = net.flat_params.clone()
    loss4 = step(xd, target_g).cpu().numpy()
    torch.cuda.synchronize()
    assert torch.equal(net.flat_params, flat_before)
    grads = net.flat_grads.clone()
    tape = cap["tape"]
    t_gpu = time.time() - t0

    # ---- the oracle's run of the same step, with every conv layer's tensors captured on the GPU (NHWC) ----
    t0 = time.time()
    lo = {}

    def dhead_of(head):
        lo.update(O.loss(head.detach().numpy(), target_o, M, N, want_grad=True))
        return torch.from_numpy(lo["grad"])

    head_ref, trace = oracle_conv_trace(ref, x, dhead_of, DEV, skip=("head.conv",))      # (the 1x1 head has its own kernel: chain check)
    t_cpu = time.time() - t0

    # ---- chain ----
    close(cap["head"].cpu(), head_ref, 1e-4)                                         # north_star: heatmap values within 1e-4
    for i, k in enumerate(("total", "hm", "offset", "embedding")):
        assert abs(loss4[i] - lo[k]) <= 1e-4 * max(abs(lo[k]), 1e-3), (k, loss4[i], lo[k])
    close(cap["dhead"].cpu(), torch.from_numpy(lo["grad"]), 1e-4)
    bn_ref = dict(ref.named_buffers())
    for name, b in net.named_buffers():
        if b.dtype == torch.long:
            assert int(b) == int(bn_ref[name]) == 1, name
        else:
            close(b.cpu(), bn_ref[name], 1e-5)                                           # running statistics after one step
    # batch statistics of every BatchNorm vs fp64 statistics of the ORACLE's conv outputs
    stats_err = 0.0
    module_name = {id(m): n for n, m in net.named_modules()}
    bn_stats = [("adpater.0", tape["stem"][2], tape["stem"][3])]
    for (blk, xin, hw, d1, c1, a1, m1, i1, d2, c2, m2, i2, out, dd, cd, md, idd, msk) in tape["blocks"]:
        pre = module_name[id(blk)]
        bn_stats += [(f"{pre}.conv1", m1, i1), (f"{pre}.conv2", m2, i2)] + ([(f"{pre}.downsample.0", md, idd)] if dd is not None else [])
    for (fpn, sc_t, hw, dl, t, dc, c, mf, if_, fn) in tape["fpn"]:
        bn_stats.append((f"{module_name[id(fpn)]}.conv.0", mf, if_))
    assert len(bn_stats) == 1 + 32 + 3 + 3
    for conv_name, mean, invstd in bn_stats:
        y = trace[conv_name]["y"].double()
        mu = y.mean((0, 1, 2)); var = y.var((0, 1, 2), unbiased=False)
        e_m = float((mean.double() - mu).abs().max() / (y.abs().max()))
        e_i = float(((invstd.double() - 1.0 / torch.sqrt(var + 1e-5)).abs() * torch.sqrt(var + 1e-5)).max())
        stats_err = max(stats_err, e_m, e_i)
        assert e_m <= 1e-5 and e_i <= 1e-5, (conv_name, e_m, e_i)

    # ---- per kernel family, on the oracle's own per-layer operands ----
    params = dict(ref.named_parameters())
    fam = {"fwd": {}, "dgrad": {}, "wgrad": {}}
    truthed = set()
    report = []
    for name, tr in trace.items():
        conv = tr["module"]
        k, stride, pad = conv.kernel_size[0], conv.stride[0], conv.padding[0]
        xl, yl, dyl, dxl = tr["x"], tr["y"], tr["dy"], tr["dx"]
        w_cpu = conv.weight.detach()
        wd = w_cpu.permute(0, 2, 3, 1).contiguous().to(DEV)                              # [Cout][R][S][Cin]
        cout, cin = w_cpu.shape[:2]
        if name == "adpater.0":
            d = make_desc(L, B, S, S, 3, 64, 7, 2, 3)
            y = torch.empty_like(yl)
            mean = torch.empty(64, device=DEV); invstd = torch.empty(64, device=DEV)
            ws = torch.empty(max(lib.sd_conv2d_stem_fwd_bn_stats_workspace_bytes(C.byref(d)), 256), dtype=torch.uint8, device=DEV)
            L.check(lib.sd_conv2d_stem_fwd_bn_stats(xd.data_ptr(), wd.data_ptr(), y.data_ptr(), C.byref(d), 1e-5, 0.1, 0, 0, mean.data_ptr(),
                                                    invstd.data_ptr(), ws.data_ptr(), ws.numel(), L.stream()), "stem fwd")
            e_f = _rel(y, yl)
            dw = torch.empty_like(wd)
            ws = torch.empty(max(lib.sd_conv2d_stem_wgrad_workspace_bytes(C.byref(d)), 256), dtype=torch.uint8, device=DEV)
            L.check(lib.sd_conv2d_stem_wgrad(dyl.data_ptr(), xd.data_ptr(), dw.data_ptr(), C.byref(d), 0, ws.data_ptr(), ws.numel(), L.stream()), "stem wgrad")
            e_w = _rel(dw.permute(0, 3, 1, 2).cpu(), params[name + ".weight"].grad)
            names = ("k_stem_fwd", "-", "k_stem_wgrad2")
            e_d = None
        else:
            _, Hi, Wi, _ = xl.shape
            d = make_desc(L, B, Hi, Wi, cin, cout, k, stride, pad)
            names = _names(lib, d)
            if name in EXPECTED_KERNELS:
                assert names == EXPECTED_KERNELS[name], (name, names)
            y = torch.empty_like(yl)
            if conv.bias is None:            # training entry of every conv that feeds a BatchNorm: statistics from the accumulators
                mean = torch.empty(cout, device=DEV); invstd = torch.empty(cout, device=DEV)
                ws = torch.empty(max(lib.sd_conv2d_fwd_bn_stats_workspace_bytes(C.byref(d)), 256), dtype=torch.uint8, device=DEV)
                L.check(lib.sd_conv2d_fwd_bn_stats(xl.data_ptr(), wd.data_ptr(), y.data_ptr(), C.byref(d), 1e-5, 0.1, 0, 0, mean.data_ptr(),
                                                   invstd.data_ptr(), ws.data_ptr(), ws.numel(), L.stream()), name)
            else:                            # biased 1x1 convs (up1, FPN laterals): bias in the epilogue
                bd = conv.bias.detach().to(DEV)
                L.check(lib.sd_conv2d_fwd(xl.data_ptr(), wd.data_ptr(), y.data_ptr(), C.byref(d), 0, bd.data_ptr(), 0, 0, 0, 0, 0, L.stream()), name)
            e_f = _rel(y, yl)
            wt = torch.empty(wd.numel(), device=DEV)
            L.check(lib.sd_conv2d_transpose_weights(wd.data_ptr(), wt.data_ptr(), cout, k * k, cin, L.stream()))
            dx = torch.empty_like(xl)
            L.check(lib.sd_conv2d_dgrad(dyl.data_ptr(), wt.data_ptr(), dx.data_ptr(), C.byref(d), 0, L.stream()), name)
            e_d = _rel(dx, dxl)
            dw = torch.empty_like(wd)
            ws = torch.empty(max(lib.sd_conv2d_wgrad_workspace_bytes(C.byref(d)), 256), dtype=torch.uint8, device=DEV)
            L.check(lib.sd_conv2d_wgrad(dyl.data_ptr(), xl.data_ptr(), dw.data_ptr(), C.byref(d), 0, ws.data_ptr(), ws.numel(), L.stream()), name)
            dw_nchw = dw.permute(0, 3, 1, 2).cpu()
            e_w = _rel(dw_nchw, params[name + ".weight"].grad)
        torch.cuda.synchronize()
        # fp64 yardstick, one layer per weight-gradient kernel family: the HIP sum over B*Ho*Wo ~ 10^5..10^6 products must be as close
        # to the truth as the CPU oracle's own fp32 sum (factor 2 + an fp32 rounding of the result)
        e_w_truth = e_w_cpu = None
        if names[2] not in truthed and name != "adpater.0":
            truthed.add(names[2])
            x64 = xl.permute(0, 3, 1, 2).cpu().double(); dy64 = dyl.permute(0, 3, 1, 2).cpu().double()
            truth = torch.nn.grad.conv2d_weight(x64, w_cpu.shape, dy64, stride, pad)
            sc = truth.abs().max().item()
            e_w_truth = (dw_nchw.double() - truth).abs().max().item() / sc
            e_w_cpu = (params[name + ".weight"].grad.double() - truth).abs().max().item() / sc
            assert e_w_truth <= 2 * e_w_cpu + 2e-6, (name, names[2], e_w_truth, e_w_cpu)
        tr["names"] = names
        report.append((name, names, e_f, e_d, e_w, e_w_truth, e_w_cpu))
        fam["fwd"].setdefault(names[0], []).append(e_f)
        if e_d is not None:
            fam["dgrad"].setdefault(names[1], []).append(e_d)
        fam["wgrad"].setdefault(names[2], []).append(e_w)
        del tr["x"], tr["y"], tr["dy"], tr["dx"]

    print(f"\n[bs=64 512x512 parity] HIP step {t_gpu:.1f} s, oracle step + capture {t_cpu:.1f} s, BatchNorm statistics worst {stats_err:.2e}")
    for phase in ("fwd", "dgrad", "wgrad"):
        for kname, errs in sorted(fam[phase].items()):
            print(f"  {phase:6s} {kname:34s} layers {len(errs):2d}  worst rel err {max(errs):.2e}  median {float(np.median(errs)):.2e}")
    for (name, names, e_f, e_d, e_w, e_t, e_c) in report:
        if e_t is not None:
            print(f"  fp64 yardstick {name:18s} {names[2]:22s} HIP {e_t:.2e}  CPU oracle {e_c:.2e}")
    # same operands, fp32 on both sides: only the summation order differs.  Reduction lengths: forward / data-gradient <= 4608
    # products, weight gradient B*Ho*Wo = 16k .. 1M products (the CPU oracle's own error vs fp64 is printed above)
    for kname, errs in fam["fwd"].items():
        assert max(errs) <= 2e-5, ("fwd", kname, max(errs))
    for kname, errs in fam["dgrad"].items():
        assert max(errs) <= 2e-5, ("dgrad", kname, max(errs))
    for kname, errs in fam["wgrad"].items():
        assert max(errs) <= 2e-4, ("wgrad", kname, max(errs))
    for must in ("k_conv3x3_c64_rows_f32", "k_conv3x3_patch<128, false>", "k_conv3x3_patch<64, false>"):
        assert must in fam["fwd"] and must in fam["dgrad"], must
    assert "k_wgrad3x3_ring2" in fam["wgrad"]

    # ---- chain, parameter gradients of the real step.  Yardstick: an fp64 run of the oracle (a random-init network amplifies fp32
    # rounding through 44 BatchNorm-coupled layers; what can be asked is that the HIP gradients are as close to the fp64 truth as
    # the fp32 CPU oracle's own).  bs=64 averages single ReLU-mask flips away, so this is held PER TENSOR, grouped per kernel family ----
    t0 = time.time()
    head64 = ref64(x.double())
    d64 = O.loss(head64.detach().float().numpy(), target_o, M, N, want_grad=True)["grad"]
    head64.backward(torch.from_numpy(d64).double())
    g64 = dict(ref64.named_parameters())
    t_64 = time.time() - t0
    worst = {}
    for name, p in net.named_parameters():
        off, n = net._flat_off[id(p)]
        gg = grads[off:off + n]
        gg = (gg.view(p.shape[0], p.shape[2], p.shape[3], p.shape[1]).permute(0, 3, 1, 2) if p.dim() == 4 else gg.view(p.shape)).cpu().double()
        truth = g64[name].grad
        sc = truth.abs().max().item() + 1e-300
        e_gpu = (gg - truth).abs().max().item() / sc
        e_cpu = (params[name].grad.double() - truth).abs().max().item() / sc
        cos = float(torch.dot(gg.flatten(), truth.flatten()) / (gg.norm() * truth.norm() + 1e-300))
        layer = name.rsplit(".", 1)[0]
        key = trace[layer]["names"][2] if layer in trace and "names" in trace[layer] else ("BatchNorm / bias" if p.dim() == 1 else "k_head_wgrad")
        worst.setdefault(key, []).append((e_gpu, e_cpu, cos, name))
    print(f"  step gradients vs the fp64 oracle run ({t_64:.0f} s): worst / median relative error per tensor, HIP | fp32 CPU oracle")
    for key, rows in sorted(worst.items()):
        eg, ec = np.array([r[0] for r in rows]), np.array([r[1] for r in rows])
        c = min(r[2] for r in rows)
        print(f"    {key:24s} tensors {len(rows):3d}  HIP {eg.max():.2e} / {np.median(eg):.2e} | CPU {ec.max():.2e} / {np.median(ec):.2e}"
              f"  min cosine {c:.6f}  (worst: {max(rows)[3]})")
        assert eg.max() <= 2 * ec.max() + 1e-4 and np.median(eg) <= 2 * np.median(ec) + 2e-5 and c >= 0.9998, (key, eg.max(), ec.max(), c)


def test_stress_geometry_bf16_forward_bs16_1024_vs_autocast_oracle():
    """BASELINE configs[4] geometry (1024x1024, 8 labels / 8 parts, bs=16 -- the batch bench.py times): eval forward on the bf16
    backbone (two-group 3x3 kernel on column strips, row-stream layer1, fused stem + pool, MFMA head) against the oracle under
    torch.autocast(cpu, bfloat16), with the oracle's fp32 forward as the truth: the hand-written path must be at least as close
    to fp32 as autocast is (fp32 epilogues), on the maximum AND on the rms of the head tensor."""
    from structuredetector_amd import _lib as L
    from structuredetector_amd.model import Network
    lib = L.lib()
    B, S, M, N = 16, 1024, 8, 8
    torch.set_num_threads(min(16, os.cpu_count() or 1))
    ref = O.build_reference_network(M, N, seed=41).eval()
    args = Namespace(labels={f"l{i}": i for i in range(M)}, parts={f"p{i}": i for i in range(N)}, fpn_depth=128, use_amp=True)
    net = Network(args, pretrained=False, raw_output=True)
    net.load_state_dict(ref.state_dict())
    net = net.to(DEV).eval()
    # running statistics of a trained network are not (0, 1): give every BatchNorm plausible ones so that the folded affines matter
    g = torch.Generator().manual_seed(1024)
    with torch.no_grad():
        for m in ref.modules():
            if isinstance(m, torch.nn.BatchNorm2d):
                m.running_mean.copy_(0.1 * torch.randn(m.running_mean.shape, generator=g))
                m.running_var.copy_(0.5 + torch.rand(m.running_var.shape, generator=g))
    net.load_state_dict(ref.state_dict())
    x = torch.randn(B, 3, S, S, generator=g)
    # the kernels this geometry must engage
    for (a, want) in (((B, 256, 256, 64, 64, 3, 1, 1), "k_conv3x3_c64_rows16_bf16"), ((B, 128, 128, 128, 128, 3, 1, 1), "k_conv3x3_bf16_pp"),
                      ((B, 64, 64, 256, 256, 3, 1, 1), "k_conv3x3_bf16_pp"), ((B, 256, 256, 128, 128, 3, 1, 1), "k_conv3x3_bf16_pp")):
        d = make_desc(L, *a)
        assert lib.sd_conv2d_kernel_name(C.byref(d), 16).decode() == want, (a, lib.sd_conv2d_kernel_name(C.byref(d), 16).decode())
    t0 = time.time()
    with torch.no_grad():
        want32 = torch.cat([ref(x[i:i + 4]) for i in range(0, B, 4)])
        with torch.autocast(device_type="cpu", dtype=torch.bfloat16):
            want16 = torch.cat([ref(x[i:i + 4]).float() for i in range(0, B, 4)])
        got = net(x.to(DEV)).cpu()
    assert got.shape == want32.shape == (B, M + N + 4, S // 4, S // 4) and got.dtype == torch.float32
    scale = want32.abs().max().item()
    e_max, a_max = (got - want32).abs().max().item() / scale, (want16 - want32).abs().max().item() / scale
    e_rms, a_rms = (got - want32).pow(2).mean().sqrt().item() / scale, (want16 - want32).pow(2).mean().sqrt().item() / scale
    print(f"\n[bs=16 1024x1024 bf16 forward] vs fp32 oracle: max {e_max:.3e} (autocast {a_max:.3e}), rms {e_rms:.3e} (autocast {a_rms:.3e}); {time.time() - t0:.0f} s")
    assert e_max <= 1.5 * a_max and e_rms <= 1.25 * a_rms, (e_max, a_max, e_rms, a_rms)
    # per image: no image of the batch is an outlier (a wrong strip / row-range would hit some images only)
    per = (got - want32).flatten(1).pow(2).mean(1).sqrt() / scale
    assert per.max().item() <= 2.0 * per.median().item(), per.tolist()
