#!/usr/bin/env python3
"""Error maps of k_conv3x3_c64_rows16_bf16 against torch on a small problem (bs 2, 16 x 128): per row / 16-pixel group / 8-channel group /
image for the plain forward, the residual forward, the statistics forward and the data-gradients, a run-to-run determinism check against the
32x32x16 kernel, and the spatial map of the second launch.  The tool the AGPR-copy hazard of round 5 was bisected with
(`profiles/r05_rows16_agpr_copy_hazard.txt`): the wrong sums showed only from the SECOND launch of a process on and moved between runs."""
import ctypes as C, sys
sys.path.insert(0, str(__import__('pathlib').Path(__file__).resolve().parent.parent))
import torch, torch.nn.functional as F
from structuredetector_amd import _lib as L
lib = L.lib(); dev = "cuda"
B, H, W = 2, 16, 128
L.check(lib.sd_set_option(b"conv_rows64_min_units", 1)); L.check(lib.sd_set_option(b"conv_fwd_split_k", 0))
d = L.ConvDesc()
d.B, d.Hi, d.Wi, d.Cin, d.Cout, d.R, d.S, d.stride, d.pad = B, H, W, 64, 64, 3, 3, 1, 1
d.Ho, d.Wo = H, W
g = torch.Generator().manual_seed(1)
x = torch.randn(B, 64, H, W, generator=g).bfloat16().float()
w = (torch.randn(64, 64, 3, 3, generator=g) / 24).bfloat16().float()
res = torch.randn(B, 64, H, W, generator=g).bfloat16().float()
xd = x.permute(0, 2, 3, 1).contiguous().to(dev).bfloat16(); wd = w.permute(0, 2, 3, 1).contiguous().to(dev).bfloat16()
rd = res.permute(0, 2, 3, 1).contiguous().to(dev).bfloat16()
y = torch.empty(B, H, W, 64, dtype=torch.bfloat16, device=dev)
ref = F.conv2d(x, w, None, 1, 1)
for name, r_ in (("plain", None), ("res", rd)):
    y.zero_()
    L.check(lib.sd_conv2d_fwd_bf16(xd.data_ptr(), wd.data_ptr(), y.data_ptr(), C.byref(d), 0, 0, r_.data_ptr() if r_ is not None else 0, 0, 0, 0, 0, L.stream()))
    got = y.float().cpu().permute(0, 3, 1, 2)
    want = ref + (res if r_ is not None else 0)
    err = (got - want).abs()
    print(name, "max err", err.max().item(), "kernel", lib.sd_conv2d_kernel_name(C.byref(d), 16).decode())
    print("  per row  :", [round(v, 2) for v in err.amax((0, 1, 3)).tolist()])
    print("  per x/16 :", [round(v, 2) for v in err.amax((0, 1, 2)).reshape(8, 16).amax(1).tolist()])
    print("  per ch/8 :", [round(v, 2) for v in err.amax((0, 2, 3)).reshape(8, 8).amax(1).tolist()])
mean = torch.empty(64, device=dev); invstd = torch.empty(64, device=dev); rm = torch.zeros(64, device=dev); rv = torch.ones(64, device=dev)
ws = torch.empty(lib.sd_conv2d_fwd_bf16_bn_stats_workspace_bytes(C.byref(d)), dtype=torch.uint8, device=dev)
for it in range(2):
    y.zero_()
    L.check(lib.sd_conv2d_fwd_bf16_bn_stats(xd.data_ptr(), wd.data_ptr(), y.data_ptr(), C.byref(d), 1e-5, 0.1, rm.data_ptr(), rv.data_ptr(),
                                            mean.data_ptr(), invstd.data_ptr(), ws.data_ptr(), ws.numel(), L.stream()))
    got = y.float().cpu().permute(0, 3, 1, 2)
    err = (got - ref).abs()
    print("stats max err", err.max().item())
    print("  per row  :", [round(v, 2) for v in err.amax((0, 1, 3)).tolist()])
    print("  per x/16 :", [round(v, 2) for v in err.amax((0, 1, 2)).reshape(8, 16).amax(1).tolist()])
    print("  per ch/8 :", [round(v, 2) for v in err.amax((0, 2, 3)).reshape(8, 8).amax(1).tolist()])
    print("  per image:", [round(v, 2) for v in err.amax((1, 2, 3)).tolist()])
    print("  mean err", (mean.cpu() - got.double().mean((0, 2, 3)).float()).abs().max().item())
dy = torch.randn(B, 64, H, W, generator=g).bfloat16().float()
wt = w.permute(1, 2, 3, 0).contiguous().to(dev).to(torch.bfloat16)
dyd = dy.permute(0, 2, 3, 1).contiguous().to(dev).bfloat16()
xg = x.clone().requires_grad_(True)
F.conv2d(xg, w, None, 1, 1).backward(dy)
dx = torch.empty(B, H, W, 64, dtype=torch.bfloat16, device=dev)
for name, r_, mode in (("dgrad plain", None, 0), ("dgrad res", rd, 1), ("dgrad plain again", None, 0)):
    dx.zero_()
    L.check(lib.sd_conv2d_dgrad_bf16(dyd.data_ptr(), wt.data_ptr(), dx.data_ptr(), C.byref(d), r_.data_ptr() if r_ is not None else 0, mode, L.stream()))
    got = dx.float().cpu().permute(0, 3, 1, 2)
    want = xg.grad + (res if r_ is not None else 0)
    err = (got - want).abs()
    print(name, "max err", err.max().item(), lib.sd_conv2d_kernel_name(C.byref(d), 17).decode())
    print("  per row  :", [round(v, 2) for v in err.amax((0, 1, 3)).tolist()])
    print("  per x/16 :", [round(v, 2) for v in err.amax((0, 1, 2)).reshape(8, 16).amax(1).tolist()])
    print("  per ch/8 :", [round(v, 2) for v in err.amax((0, 2, 3)).reshape(8, 8).amax(1).tolist()])
    print("  per image:", [round(v, 2) for v in err.amax((1, 2, 3)).tolist()])
print("---- determinism")
def run_fwd(xin, win):
    y.zero_()
    L.check(lib.sd_conv2d_fwd_bf16(xin.data_ptr(), win.data_ptr(), y.data_ptr(), C.byref(d), 0, 0, 0, 0, 0, 0, 0, L.stream()))
    return y.clone()
def run_dg(xin, win):
    dx.zero_()
    L.check(lib.sd_conv2d_dgrad_bf16(xin.data_ptr(), win.data_ptr(), dx.data_ptr(), C.byref(d), 0, 0, L.stream()))
    return dx.clone()
for nm, fn, a_, b_ in (("fwd(x,w)", run_fwd, xd, wd), ("fwd(dy,w)", run_fwd, dyd, wd), ("dgrad(dy,wt)", run_dg, dyd, wt), ("dgrad(x,wt)", run_dg, xd, wt)):
    outs = [fn(a_, b_) for _ in range(4)]
    L.check(lib.sd_set_option(b"conv_rows16", 0))
    base = fn(a_, b_)
    L.check(lib.sd_set_option(b"conv_rows16", 1))
    print(nm, "same across runs:", [bool(torch.equal(outs[0], o_)) for o_ in outs[1:]], "max diff vs 32x32x16 kernel:", [(o_.float() - base.float()).abs().max().item() for o_ in outs])
print("---- error map of the second fwd(x, w) launch vs torch: rows x (x / 8), max over channels, image 0 then 1")
o1 = run_fwd(xd, wd); o2 = run_fwd(xd, wd)
got = o2.float().cpu().permute(0, 3, 1, 2)
err = (got - ref).abs()
for b_ in range(B):
    print("image", b_)
    m = err[b_].amax(0).reshape(H, W // 8, 8).amax(2)
    for r_ in range(H):
        print("  row %2d: " % r_ + " ".join("%4.1f" % v for v in m[r_].tolist()))
m2 = err.amax((0, 2, 3))
print("per channel:", " ".join("%3.1f" % v for v in m2.tolist()))
