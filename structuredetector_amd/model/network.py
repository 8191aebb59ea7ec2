"""`Network`: ResNet-34 trunk + 3-level FPN + 1x1 head on hand-written HIP kernels.

Mirrors `Network` / `Fpn` / `Head` of src/sdnet/model/network.py:6-87 and the torchvision
resnet34 pieces it borrows (conv1/bn1/relu/maxpool/layer1-4; BasicBlock [3,4,6,3]):
same constructor signature, same `forward(x) -> dict` of four channel-slice views (or the raw
tensor with `raw_output=True`), same `save()`, and the same state_dict key schema (including the
`adpater` spelling) so reference `.pth` checkpoints load unchanged.

Mechanism (MI355X-first, not a module-by-module translation):
  * every learnable tensor lives in ONE flat fp32 device buffer (`flat_params`), gradients in a
    second one (`flat_grads`): the optimizer is a single fused Adam launch and the data-parallel
    exchange is a single RCCL all-reduce over the flat gradient buffer;
  * activations are NHWC, conv weights [Cout][R][S][Cin] (the parameters are exposed with OIHW
    *logical* shape and channels_last strides, so state_dicts stay interchangeable);
  * forward and backward are explicit kernel schedules (`_Engine`) behind one autograd node;
    no torch.nn.functional / MIOpen call is made anywhere.
"""
from __future__ import annotations

import ctypes as C
import math
import warnings

import torch
import torch.nn as nn

from .. import _lib as L
from ..utils import trace as T

BN_EPS = 1e-5
BN_MOMENTUM = 0.1


# ------------------------------------------------------------------------------------------
# parameter containers (no compute): they only give the state_dict its reference key names
# ------------------------------------------------------------------------------------------
class ConvParams(nn.Module):
    def __init__(self, cin, cout, k, stride=1, pad=0, bias=False):
        super().__init__()
        self.cin, self.cout, self.k, self.stride, self.pad = cin, cout, k, stride, pad
        w = torch.empty((cout, cin, k, k), memory_format=torch.channels_last)
        self.weight = nn.Parameter(w)
        self.bias = nn.Parameter(torch.zeros(cout)) if bias else None


class BNParams(nn.Module):
    def __init__(self, c):
        super().__init__()
        self.c = c
        self.weight = nn.Parameter(torch.ones(c))
        self.bias = nn.Parameter(torch.zeros(c))
        self.register_buffer("running_mean", torch.zeros(c))
        self.register_buffer("running_var", torch.ones(c))
        self.register_buffer("num_batches_tracked", torch.tensor(0, dtype=torch.long))


class Slot(nn.Module):
    """Parameter-less position (ReLU / MaxPool / Upsample) kept so Sequential indices match the reference."""


class BasicBlock(nn.Module):
    def __init__(self, cin, cout, stride):
        super().__init__()
        self.conv1 = ConvParams(cin, cout, 3, stride, 1)
        self.bn1 = BNParams(cout)
        self.relu = Slot()
        self.conv2 = ConvParams(cout, cout, 3, 1, 1)
        self.bn2 = BNParams(cout)
        self.downsample = None
        if stride != 1 or cin != cout:
            self.downsample = nn.Sequential(ConvParams(cin, cout, 1, stride, 0), BNParams(cout))


class Fpn(nn.Module):
    """network.py:6-19: conv3x3+BN+ReLU( upsample2x(input) + lateral1x1(shortcut) )."""

    def __init__(self, in_channels, out_channels):
        super().__init__()
        self.up = Slot()
        self.lateral = ConvParams(in_channels, out_channels, 1, bias=True)
        self.conv = nn.Sequential(ConvParams(out_channels, out_channels, 3, 1, 1), BNParams(out_channels), Slot())


class Head(nn.Module):
    """network.py:22-29."""

    def __init__(self, in_channels, out_channels):
        super().__init__()
        self.conv = ConvParams(in_channels, out_channels, 1, bias=True)


BACKBONE_FILE_GLOB = "resnet34-*.pth"          # torchvision ResNet34_Weights.DEFAULT = IMAGENET1K_V1 = resnet34-b627a593.pth


def find_backbone_weights(args=None):
    """Local lookup of the ImageNet ResNet-34 checkpoint `pretrained=True` asks for (network.py:41), in this order:
    `args.backbone_weights`, $SDNET_BACKBONE_WEIGHTS (both must exist when given), then `resnet34-*.pth` in the directories
    torchvision / torch.hub would have cached it in ($TORCH_HOME/hub/checkpoints, $XDG_CACHE_HOME/torch/hub/checkpoints,
    ~/.cache/torch/hub/checkpoints).  Returns a Path or None; never downloads."""
    import os
    from pathlib import Path
    for given in (getattr(args, "backbone_weights", None), os.environ.get("SDNET_BACKBONE_WEIGHTS")):
        if given:
            path = Path(given).expanduser()
            if not path.is_file():
                raise L.SdError(f"backbone weights file not found: {path}")
            return path
    homes = []
    if os.environ.get("TORCH_HOME"):
        homes.append(Path(os.environ["TORCH_HOME"]))
    homes.append(Path(os.environ.get("XDG_CACHE_HOME", "~/.cache")).expanduser() / "torch")
    for home in homes:
        hits = sorted((home / "hub" / "checkpoints").glob(BACKBONE_FILE_GLOB))
        if hits:
            return hits[0]
    return None


def _layer(cin, cout, n, stride):
    return nn.Sequential(*[BasicBlock(cin if i == 0 else cout, cout, stride if i == 0 else 1) for i in range(n)])


# ------------------------------------------------------------------------------------------
# kernel schedule
# ------------------------------------------------------------------------------------------
def _desc(B, Hi, Wi, conv: ConvParams):
    d = L.ConvDesc()
    d.B, d.Hi, d.Wi, d.Cin, d.Cout, d.R, d.S, d.stride, d.pad = B, Hi, Wi, conv.cin, conv.cout, conv.k, conv.k, conv.stride, conv.pad
    d.Ho = (Hi + 2 * conv.pad - conv.k) // conv.stride + 1
    d.Wo = (Wi + 2 * conv.pad - conv.k) // conv.stride + 1
    return d


def _ptr(t):
    return 0 if t is None else t.data_ptr()


class _Engine:
    """Explicit forward / backward schedules over the C ABI.  Tensors named *_nhwc are
    (B, H, W, C) contiguous; `tape` keeps what the backward needs."""

    def __init__(self, net: "Network"):
        self.net = net
        self.lib = L.lib()
        self.prof = None      # list -> every conv launch appends (kernel, algorithmic flops, start event, end event, phase)
        self.prof_shapes = None   # list -> the geometry of every labelled launch, in the order of `prof` (tools/step_layers.py)
        self._nbt = []        # num_batches_tracked counters touched by the running forward (bumped with one launch)
        # Optional: weight-gradient GEMMs on a side stream (they only depend on the layer's output gradient), free-running
        # beside the data-gradient chain.  Measured +3.5 % images/s at bs=64 in round 3 and -3 % in round 4 (the row-ring weight gradient's
        # 512-thread block owns 152 KB of LDS: no conv block fits beside it); OFF by default because concurrent kernels
        # stretch each other's durations and the per-kernel roofline numbers of bench.py / rocprofv3 stop describing a
        # kernel running alone.  (Chaining dgrad -> wgrad -> dgrad across the streams so that only the BatchNorm passes
        # overlap was measured too: slower than no overlap, the HBM-bound passes slow the wgrad they run under.)
        self.overlap_wgrad = False
        self._side = None
        # BatchNorm-backward reduction inside the data-gradient epilogue (sd_conv2d_dgrad_bn_reduce).  Measured at bs=64: with the
        # first (4-byte) epilogue it cost the conv kernels 4.0 ms to save 2.2 ms of reduce passes; with the 16-byte epilogue
        # (float4 reads of the BatchNorm input, mask bytes) it is exactly neutral (815.7 vs 815.4 img/s): the reduce passes it
        # removes run at HBM speed anyway.  Off by default; the path is kept and tested (tests/test_gpu_network.py).
        self.fuse_bn_bwd = False
        self._wt_plan, self._wt_flat, self._wt_valid = None, {}, None      # transposed data-gradient weights (see _transpose_all)
        self._head_ok = {}
        self.stem_ring = True              # mixed-precision step: bf16 stem gradient + the row-ring stem weight gradient (False: fp32 gradient + k_stem_wgrad_bf16; A/B)
        self.fuse_head = True              # bf16 inference: the 1x1 head in the epilogue of the last FPN conv where it takes the two-group kernel (False: A/B, tests)
        self.small_batch_kernel = True     # eval forward: sd_conv2d_fwd_sb where the 128-row tile grid cannot fill the chip (False: A/B)

    def _kname(self, d, which):
        """device kernel the C ABI will launch for this conv (profiling label; same names as the rocprofv3 kernel trace)"""
        if self.prof is None:
            return ""
        if self.prof_shapes is not None:
            self.prof_shapes.append((d.B, d.Hi, d.Wi, d.Cin, d.Cout, d.R, d.stride, d.Ho, d.Wo))
        return self.lib.sd_conv2d_kernel_name(C.byref(d), which).decode()

    def _timed(self, kind, flops, fn, phase="fwd"):
        if self.prof is None:
            return fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()           # torch's current stream == the stream handed to the C ABI (L.stream())
        fn()
        e1.record()
        self.prof.append((kind, flops, e0, e1, phase))

    # ---- helpers -------------------------------------------------------------------------
    def _ws(self, nbytes, dev):
        return L.workspace(max(int(nbytes), 256), dev)

    # ---- mixed precision (`--amp`, trainer.py:115-121): bf16 activations / conv weights, fp32 accumulation and master weights ----
    def _to_bf16(self, t):
        out = torch.empty(t.shape, dtype=torch.bfloat16, device=t.device)
        L.check(self.lib.sd_cast_f32_to_bf16(t.data_ptr(), out.data_ptr(), t.numel(), L.stream()), "sd_cast_f32_to_bf16")
        return out

    def _to_f32(self, t):
        out = torch.empty(t.shape, dtype=torch.float32, device=t.device)
        L.check(self.lib.sd_cast_bf16_to_f32(t.data_ptr(), out.data_ptr(), t.numel(), L.stream()), "sd_cast_bf16_to_f32")
        return out

    def refresh_bf16_weights(self):
        """One cast of the whole flat fp32 parameter buffer per step; every conv weight is then a view at its fp32 offset."""
        net = self.net
        if net.flat_params_bf16 is None:
            net.flat_params_bf16 = torch.empty(net.flat_params.numel(), dtype=torch.bfloat16, device=net.flat_params.device)
        L.check(self.lib.sd_cast_f32_to_bf16(net.flat_params.data_ptr(), net.flat_params_bf16.data_ptr(), net.flat_params.numel(), L.stream()),
                "sd_cast_f32_to_bf16")

    def _w16(self, conv):
        off, n = self.net._flat_off[id(conv.weight)]
        return self.net.flat_params_bf16[off:off + n]                 # physical [Cout][R][S][Cin] order; off % 8 == 0 (32-byte slots): 16-byte aligned

    def _conv_sb(self, x, w, y, d, scale, shift, res, res_up2, relu, bf16):
        """Small-batch inference conv: one launch, split-K combined inside it (sd_conv2d_fwd_sb); the arrival tickets live in a
        zeroed per-stream state buffer that every launch leaves zero."""
        nws = self.lib.sd_conv2d_fwd_sb_workspace_bytes(C.byref(d), bf16)
        ws = self._ws(nws, x.device) if nws else None
        st = L.zero_state(self.lib.sd_conv2d_fwd_sb_state_bytes(C.byref(d), bf16), x.device) if nws else None
        L.check(self.lib.sd_conv2d_fwd_sb(x.data_ptr(), w.data_ptr(), y.data_ptr(), C.byref(d), _ptr(scale), _ptr(shift), _ptr(res), int(res_up2),
                                          int(relu), bf16, _ptr(ws), ws.numel() if nws else 0, _ptr(st), st.numel() if nws else 0, L.stream()),
                "sd_conv2d_fwd_sb")

    def conv(self, x, conv, B, Hi, Wi, scale=None, shift=None, res=None, res_up2=False, relu=False, amp=False, sb=False):
        d = _desc(B, Hi, Wi, conv)
        if amp:
            y = torch.empty((B, d.Ho, d.Wo, conv.cout), dtype=torch.bfloat16, device=x.device)
            flops = 2.0 * B * d.Ho * d.Wo * conv.cout * conv.cin * conv.k * conv.k
            self._timed("bf16:" + self._kname(d, 16), flops, lambda: L.check(
                self.lib.sd_conv2d_fwd_bf16(x.data_ptr(), self._w16(conv).data_ptr(), y.data_ptr(), C.byref(d), _ptr(scale), _ptr(shift),
                                            _ptr(res), int(res_up2), int(relu), 0, 0, L.stream()), "sd_conv2d_fwd_bf16"))
            return y, d
        y = torch.empty((B, d.Ho, d.Wo, conv.cout), dtype=torch.float32, device=x.device)
        flops = 2.0 * B * d.Ho * d.Wo * conv.cout * conv.cin * conv.k * conv.k
        if sb and self.small_batch_kernel and self.lib.sd_conv2d_fwd_sb_supported(C.byref(d), 0):
            self._timed("k_conv_fwd_sb<false>" if self.prof is not None else "", flops,
                        lambda: self._conv_sb(x, conv.weight, y, d, scale, shift, res, res_up2, relu, 0))
            return y, d
        nws = self.lib.sd_conv2d_fwd_workspace_bytes(C.byref(d))          # > 0 only for small batches (split-K)
        ws = self._ws(nws, x.device) if nws else None
        self._timed(self._kname(d, 0), flops, lambda: L.check(
            self.lib.sd_conv2d_fwd(x.data_ptr(), conv.weight.data_ptr(), y.data_ptr(), C.byref(d), _ptr(scale), _ptr(shift),
                                   _ptr(res), int(res_up2), int(relu), _ptr(ws), ws.numel() if nws else 0, L.stream()),
            "sd_conv2d_fwd"))
        return y, d

    def conv_stats(self, x, conv, B, Hi, Wi, bn: BNParams, update_running=True, amp=False):
        """Training forward of conv -> BatchNorm: the conv launch also produces the batch statistics of its output
        (sd_conv2d_fwd_bn_stats), so bn_train(..., stats=...) only has the apply pass left."""
        d = _desc(B, Hi, Wi, conv)
        y = torch.empty((B, d.Ho, d.Wo, conv.cout), dtype=torch.bfloat16 if amp else torch.float32, device=x.device)
        mean = torch.empty(conv.cout, dtype=torch.float32, device=x.device)
        invstd = torch.empty_like(mean)
        flops = 2.0 * B * d.Ho * d.Wo * conv.cout * conv.cin * conv.k * conv.k
        if amp:
            ws = self._ws(self.lib.sd_conv2d_fwd_bf16_bn_stats_workspace_bytes(C.byref(d)), x.device)
            self._timed("bf16:" + self._kname(d, 16), flops, lambda: L.check(
                self.lib.sd_conv2d_fwd_bf16_bn_stats(x.data_ptr(), self._w16(conv).data_ptr(), y.data_ptr(), C.byref(d), BN_EPS, BN_MOMENTUM,
                                                     bn.running_mean.data_ptr() if update_running else 0,
                                                     bn.running_var.data_ptr() if update_running else 0,
                                                     mean.data_ptr(), invstd.data_ptr(), ws.data_ptr(), ws.numel(), L.stream()),
                "sd_conv2d_fwd_bf16_bn_stats"))
            return y, d, (mean, invstd)
        ws = self._ws(self.lib.sd_conv2d_fwd_bn_stats_workspace_bytes(C.byref(d)), x.device)
        self._timed(self._kname(d, 0), flops, lambda: L.check(
            self.lib.sd_conv2d_fwd_bn_stats(x.data_ptr(), conv.weight.data_ptr(), y.data_ptr(), C.byref(d), BN_EPS, BN_MOMENTUM,
                                            bn.running_mean.data_ptr() if update_running else 0,
                                            bn.running_var.data_ptr() if update_running else 0,
                                            mean.data_ptr(), invstd.data_ptr(), ws.data_ptr(), ws.numel(), L.stream()),
            "sd_conv2d_fwd_bn_stats"))
        return y, d, (mean, invstd)

    def bn_train(self, x, bn: BNParams, res=None, relu=True, update_running=True, stats=None, want_mask=False):
        Mrows, Cc = x.numel() // x.shape[-1], x.shape[-1]
        if stats is not None:
            mean, invstd = stats
        else:
            mean = torch.empty(Cc, dtype=torch.float32, device=x.device)
            invstd = torch.empty_like(mean)
            ws = self._ws(self.lib.sd_col_reduce_workspace_bytes(Mrows, Cc), x.device)
            L.check(self.lib.sd_bn_train_stats(x.data_ptr(), Mrows, Cc, BN_EPS, BN_MOMENTUM,
                                               bn.running_mean.data_ptr() if update_running else 0,
                                               bn.running_var.data_ptr() if update_running else 0,
                                               mean.data_ptr(), invstd.data_ptr(), ws.data_ptr(), ws.numel(), L.stream()), "sd_bn_train_stats")
        y = torch.empty_like(x)
        # residual layers: the backward needs the ReLU mask of y; one byte per four elements instead of re-reading y twice
        mask = torch.empty(x.numel() // 4, dtype=torch.uint8, device=x.device) if want_mask else None
        apply = self.lib.sd_bn_apply_bf16 if x.dtype == torch.bfloat16 else self.lib.sd_bn_apply
        L.check(apply(x.data_ptr(), y.data_ptr(), Mrows, Cc, mean.data_ptr(), invstd.data_ptr(), bn.weight.data_ptr(),
                      bn.bias.data_ptr(), _ptr(res), int(relu), _ptr(mask), L.stream()), "sd_bn_apply")
        if update_running:
            self._nbt.append(bn.num_batches_tracked)
        return (y, mean, invstd, mask) if want_mask else (y, mean, invstd)

    def bn_fold(self, bn: BNParams):
        cached = self.net._folded.get(id(bn))
        if cached is not None:
            return cached
        scale = torch.empty(bn.c, dtype=torch.float32, device=bn.weight.device)
        shift = torch.empty_like(scale)
        L.check(self.lib.sd_bn_fold(bn.weight.data_ptr(), bn.bias.data_ptr(), bn.running_mean.data_ptr(), bn.running_var.data_ptr(),
                                    BN_EPS, bn.c, scale.data_ptr(), shift.data_ptr(), L.stream()), "sd_bn_fold")
        self.net._folded[id(bn)] = (scale, shift)
        return scale, shift

    # ---- forward -------------------------------------------------------------------------
    def forward(self, x, training, tape=None, amp=False):
        net, lib = self.net, self.lib
        if amp:
            if not training:
                raise L.SdError("amp=True is the mixed-precision TRAINING forward; inference uses forward_bf16")
            self.refresh_bf16_weights()
        if x.dim() != 4 or x.shape[1] != 3:
            raise L.SdError(f"Network expects (B, 3, H, W) input, got {tuple(x.shape)}")
        L.require_cuda(x)
        x = x.contiguous().float()
        B, _, H, W = x.shape
        if H % 32 or W % 32:
            raise L.SdError("input height/width must be multiples of 32 (args.py:181-186)")
        rec = tape is not None

        # stem: conv7x7/2 + BN + ReLU + maxpool3x3/2 (network.py:43-45)
        T.push("fwd:stem")
        stem, bn0 = net.adpater[0], net.adpater[1]
        d0 = _desc(B, H, W, stem)
        # mixed precision: the stem's conv output and pooled map are bf16 too (even maps: the quad form of the tail's backward); BatchNorm
        # statistics / arithmetic stay fp32
        amp_stem = bool(amp and training and d0.Ho % 2 == 0 and d0.Wo % 2 == 0)
        s0 = torch.empty((B, d0.Ho, d0.Wo, 64), dtype=torch.bfloat16 if amp_stem else torch.float32, device=x.device)
        wss = self._ws(lib.sd_conv2d_stem_fwd_workspace_bytes(C.byref(d0)), x.device)
        Hp, Wp = (d0.Ho + 2 - 3) // 2 + 1, (d0.Wo + 2 - 3) // 2 + 1
        p1 = torch.empty((B, Hp, Wp, 64), dtype=torch.bfloat16 if amp_stem else torch.float32, device=x.device)
        pidx = torch.empty((B, Hp, Wp, 64), dtype=torch.uint8, device=x.device)
        if training:
            # BatchNorm + ReLU + max-pool in one pass over the conv output: the full-resolution activation (1 GB at bs=64,
            # 512x512) is neither written nor read back; the backward recomputes what it needs from s0 and the pool indices
            m0 = torch.empty(64, dtype=torch.float32, device=x.device)
            i0 = torch.empty_like(m0)
            ws = self._ws(lib.sd_conv2d_stem_fwd_bn_stats_workspace_bytes(C.byref(d0)), x.device)
            stem_fwd = (lib.sd_conv2d_stem_fwd_bn_stats_bf16 if amp_stem else lib.sd_conv2d_stem_fwd_bn_stats_bf16mm) if amp \
                else lib.sd_conv2d_stem_fwd_bn_stats                                                           # amp: product on the bf16 MFMA
            L.check(stem_fwd(x.data_ptr(), stem.weight.data_ptr(), s0.data_ptr(), C.byref(d0), BN_EPS, BN_MOMENTUM,
                             bn0.running_mean.data_ptr(), bn0.running_var.data_ptr(), m0.data_ptr(), i0.data_ptr(),
                             ws.data_ptr(), ws.numel(), L.stream()), "sd_conv2d_stem_fwd_bn_stats")
            self._nbt.append(bn0.num_batches_tracked)
            pool_fwd = lib.sd_bn_relu_maxpool_fwd_bf16 if amp_stem else lib.sd_bn_relu_maxpool_fwd
            L.check(pool_fwd(s0.data_ptr(), B, d0.Ho, d0.Wo, 64, m0.data_ptr(), i0.data_ptr(), bn0.weight.data_ptr(),
                             bn0.bias.data_ptr(), p1.data_ptr(), pidx.data_ptr(), L.stream()), "sd_bn_relu_maxpool_fwd")
        else:
            sc, sh = self.bn_fold(bn0)
            L.check(lib.sd_conv2d_stem_fwd(x.data_ptr(), stem.weight.data_ptr(), s0.data_ptr(), C.byref(d0), sc.data_ptr(), sh.data_ptr(), 1, 0,
                                           wss.data_ptr(), wss.numel(), L.stream()), "stem")
            m0 = i0 = None
            L.check(lib.sd_maxpool3x3s2_fwd(s0.data_ptr(), p1.data_ptr(), pidx.data_ptr(), B, d0.Ho, d0.Wo, 64, L.stream()), "maxpool")
        if rec:
            tape["x"], tape["stem"], tape["amp"] = x, (d0, s0, m0, i0, pidx), bool(amp)
        if amp and not amp_stem:
            p1 = self._to_bf16(p1)          # (odd maps: fp32 stem tail) everything behind the max-pool is bf16

        T.pop()
        # trunk (network.py:47-50)
        feats, cur, Hc, Wc = [], p1, Hp, Wp
        blocks_tape = []
        for li, layer in enumerate((net.down1, net.down2, net.down3, net.down4), start=1):
            T.push(f"fwd:down{li}")
            for blk in layer:
                if training:
                    c1, d1, st1 = self.conv_stats(cur, blk.conv1, B, Hc, Wc, blk.bn1, amp=amp)
                    a1, m1, i1 = self.bn_train(c1, blk.bn1, stats=st1)
                    c2, d2, st2 = self.conv_stats(a1, blk.conv2, B, d1.Ho, d1.Wo, blk.bn2, amp=amp)
                    if blk.downsample is not None:
                        cd, dd, std_ = self.conv_stats(cur, blk.downsample[0], B, Hc, Wc, blk.downsample[1], amp=amp)
                        idt, md, idd = self.bn_train(cd, blk.downsample[1], relu=False, stats=std_)
                    else:
                        cd = dd = md = idd = None
                        idt = cur
                    out, m2, i2, msk = self.bn_train(c2, blk.bn2, res=idt, relu=True, stats=st2, want_mask=True)
                    if rec:
                        blocks_tape.append((blk, cur, (Hc, Wc), d1, c1, a1, m1, i1, d2, c2, m2, i2, out, dd, cd, md, idd, msk))
                else:
                    s1, h1 = self.bn_fold(blk.bn1)
                    a1, d1 = self.conv(cur, blk.conv1, B, Hc, Wc, scale=s1, shift=h1, relu=True, sb=True)
                    if blk.downsample is not None:
                        sd_, hd = self.bn_fold(blk.downsample[1])
                        idt, _ = self.conv(cur, blk.downsample[0], B, Hc, Wc, scale=sd_, shift=hd, sb=True)
                    else:
                        idt = cur
                    s2, h2 = self.bn_fold(blk.bn2)
                    out, _ = self.conv(a1, blk.conv2, B, d1.Ho, d1.Wo, scale=s2, shift=h2, res=idt, relu=True, sb=True)
                cur, Hc, Wc = out, d1.Ho, d1.Wo
            feats.append((cur, Hc, Wc))
            T.pop()
        (p2, H2, W2), (p3, H3, W3), (p4, H4, W4), (p5, H5, W5) = feats

        # FPN (network.py:52-55,6-19): lateral 1x1 (+bias) with the x2-upsampled coarser map added in the epilogue
        T.push("fwd:fpn")
        f, _ = self.conv(p5, net.up1, B, H5, W5, shift=net.up1.bias, amp=amp, sb=not training)
        fpn_tape = []
        for fpn, (sc_t, Hs, Ws) in ((net.up2, (p4, H4, W4)), (net.up3, (p3, H3, W3)), (net.up4, (p2, H2, W2))):
            t, dl = self.conv(sc_t, fpn.lateral, B, Hs, Ws, shift=fpn.lateral.bias, res=f, res_up2=True, amp=amp, sb=not training)
            if training:
                c, dc, stf = self.conv_stats(t, fpn.conv[0], B, Hs, Ws, fpn.conv[1], amp=amp)
                fn, mf, if_ = self.bn_train(c, fpn.conv[1], stats=stf)
                if rec:
                    fpn_tape.append((fpn, sc_t, (Hs, Ws), dl, t, dc, c, mf, if_, fn))
            else:
                sf, hf = self.bn_fold(fpn.conv[1])
                fn, _ = self.conv(t, fpn.conv[0], B, Hs, Ws, scale=sf, shift=hf, relu=True, sb=True)
            f = fn
        T.pop()

        # head (network.py:57): NHWC -> NCHW
        T.push("fwd:head")
        hc = net.head.conv
        out = torch.empty((B, hc.cout, H2, W2), dtype=torch.float32, device=x.device)
        head_fwd = lib.sd_head_fwd_bf16 if amp else lib.sd_head_fwd       # (the head's output and the loss stay fp32)
        L.check(head_fwd(f.data_ptr(), hc.weight.data_ptr(), hc.bias.data_ptr(), out.data_ptr(), B, H2 * W2, hc.cin, hc.cout,
                         L.stream()), "sd_head_fwd")
        T.pop()
        if rec:
            tape.update(blocks=blocks_tape, fpn=fpn_tape, p5=(p5, H5, W5), f1=f, B=B, hw=(H2, W2))
        if self._nbt:
            torch._foreach_add_(self._nbt, 1)
            self._nbt = []
        return out

    # ---- bf16 backbone, inference only (BASELINE stress config: "bf16 backbone + fp32 decode") -------------
    def _w_bf16(self, conv):
        """bf16 copy of a conv weight in its physical [Cout][R][S][Cin] order (cached with the folded BN affines)."""
        key = ("w", id(conv))
        w = self.net._folded.get(key)
        if w is None:
            n = conv.weight.numel()
            w = torch.empty(n, dtype=torch.bfloat16, device=conv.weight.device)
            L.check(self.lib.sd_cast_f32_to_bf16(conv.weight.data_ptr(), w.data_ptr(), n, L.stream()), "sd_cast_f32_to_bf16")
            self.net._folded[key] = w
        return w

    def _head_fusable(self, B, Hs, Ws, conv, hc):
        """sd_conv2d_fwd_bf16_head_supported, remembered per shape (a batch-1 forward is host-bound: no descriptor + ctypes call per forward)."""
        key = (B, Hs, Ws, hc.cout)
        ok = self._head_ok.get(key)
        if ok is None:
            ok = hc.cin == 128 and bool(self.lib.sd_conv2d_fwd_bf16_head_supported(C.byref(_desc(B, Hs, Ws, conv)), hc.cout))
            self._head_ok[key] = ok
        return ok

    def _head_prepared(self, hc):
        """hi / lo bf16 halves of the fp32 head weights + the padded bias for sd_conv2d_fwd_bf16_head (cached with the folded BN affines)."""
        key = ("head", id(hc))
        buf = self.net._folded.get(key)
        if buf is None:
            buf = torch.empty(self.lib.sd_head_split_bf16_bytes(), dtype=torch.uint8, device=hc.weight.device)
            L.check(self.lib.sd_head_split_bf16(hc.weight.data_ptr(), hc.bias.data_ptr(), hc.cout, buf.data_ptr(), L.stream()), "sd_head_split_bf16")
            self.net._folded[key] = buf
        return buf

    def conv_bf16(self, x, conv, B, Hi, Wi, scale=None, shift=None, res=None, res_up2=False, relu=False):
        d = _desc(B, Hi, Wi, conv)
        y = torch.empty((B, d.Ho, d.Wo, conv.cout), dtype=torch.bfloat16, device=x.device)
        if self.small_batch_kernel and self.lib.sd_conv2d_fwd_sb_supported(C.byref(d), 1):
            self._conv_sb(x, self._w_bf16(conv), y, d, scale, shift, res, res_up2, relu, 1)
            return y, d
        nws = self.lib.sd_conv2d_fwd_bf16_workspace_bytes(C.byref(d))
        ws = self._ws(nws, x.device) if nws else None
        L.check(self.lib.sd_conv2d_fwd_bf16(x.data_ptr(), self._w_bf16(conv).data_ptr(), y.data_ptr(), C.byref(d), _ptr(scale), _ptr(shift),
                                            _ptr(res), int(res_up2), int(relu), _ptr(ws), ws.numel() if nws else 0, L.stream()),
                "sd_conv2d_fwd_bf16")
        return y, d

    def forward_bf16(self, x):
        """Eval-mode forward with bf16 activations / weights and fp32 accumulation; BN folded into fp32 epilogues.
        The stem reads the fp32 image with fp32 weights and stores bf16; the head returns fp32 NCHW."""
        net, lib = self.net, self.lib
        L.require_cuda(x)
        x = x.contiguous().float()
        B, _, H, W = x.shape
        if x.dim() != 4 or x.shape[1] != 3 or H % 32 or W % 32:
            raise L.SdError("Network expects (B, 3, H, W) input with H, W multiples of 32")
        stem, bn0 = net.adpater[0], net.adpater[1]
        d0 = _desc(B, H, W, stem)
        sc, sh = self.bn_fold(bn0)
        Hc, Wc = (d0.Ho + 2 - 3) // 2 + 1, (d0.Wo + 2 - 3) // 2 + 1
        cur = torch.empty((B, Hc, Wc, 64), dtype=torch.bfloat16, device=x.device)
        if stem.k == 7 and stem.stride == 2 and stem.pad == 3 and d0.Wo <= 4096:
            # conv1 -> bn1 -> relu -> maxpool in one launch: the (B, H/2, W/2, 64) activation is never written
            L.check(lib.sd_stem_bn_relu_maxpool_fwd_bf16(x.data_ptr(), stem.weight.data_ptr(), sc.data_ptr(), sh.data_ptr(), cur.data_ptr(),
                                                         C.byref(d0), L.stream()), "stem+maxpool")
        else:
            a0 = torch.empty((B, d0.Ho, d0.Wo, 64), dtype=torch.bfloat16, device=x.device)
            wss = self._ws(lib.sd_conv2d_stem_fwd_workspace_bytes(C.byref(d0)), x.device)
            L.check(lib.sd_conv2d_stem_fwd(x.data_ptr(), stem.weight.data_ptr(), a0.data_ptr(), C.byref(d0), sc.data_ptr(), sh.data_ptr(), 1, 1,
                                           wss.data_ptr(), wss.numel(), L.stream()), "stem")
            L.check(lib.sd_maxpool3x3s2_fwd_bf16(a0.data_ptr(), cur.data_ptr(), B, d0.Ho, d0.Wo, 64, L.stream()), "maxpool")
        feats = []
        for layer in (net.down1, net.down2, net.down3, net.down4):
            for blk in layer:
                s1, h1 = self.bn_fold(blk.bn1)
                a1, d1 = self.conv_bf16(cur, blk.conv1, B, Hc, Wc, scale=s1, shift=h1, relu=True)
                if blk.downsample is not None:
                    sd_, hd = self.bn_fold(blk.downsample[1])
                    idt, _ = self.conv_bf16(cur, blk.downsample[0], B, Hc, Wc, scale=sd_, shift=hd)
                else:
                    idt = cur
                s2, h2 = self.bn_fold(blk.bn2)
                cur, _ = self.conv_bf16(a1, blk.conv2, B, d1.Ho, d1.Wo, scale=s2, shift=h2, res=idt, relu=True)
                Hc, Wc = d1.Ho, d1.Wo
            feats.append((cur, Hc, Wc))
        (p2, H2, W2), (p3, H3, W3), (p4, H4, W4), (p5, H5, W5) = feats
        f, _ = self.conv_bf16(p5, net.up1, B, H5, W5, shift=net.up1.bias)
        hc = net.head.conv
        out = torch.empty((B, hc.cout, H2, W2), dtype=torch.float32, device=x.device)
        for fpn, (sc_t, Hs, Ws) in ((net.up2, (p4, H4, W4)), (net.up3, (p3, H3, W3)), (net.up4, (p2, H2, W2))):
            t, _ = self.conv_bf16(sc_t, fpn.lateral, B, Hs, Ws, shift=fpn.lateral.bias, res=f, res_up2=True)
            sf, hf = self.bn_fold(fpn.conv[1])
            if fpn is net.up4 and self.fuse_head and self._head_fusable(B, Hs, Ws, fpn.conv[0], hc):
                # network.py:17-18 + 22-29 in one launch where the conv takes the two-group kernel: the FPN output is never stored
                dl = _desc(B, Hs, Ws, fpn.conv[0])
                rc = lib.sd_conv2d_fwd_bf16_head(t.data_ptr(), self._w_bf16(fpn.conv[0]).data_ptr(), C.byref(dl), sf.data_ptr(), hf.data_ptr(), 1,
                                                 self._head_prepared(hc).data_ptr(), hc.cout, out.data_ptr(), L.stream())
                if rc == 0:
                    return out
                if rc > 0:
                    L.check(rc, "sd_conv2d_fwd_bf16_head")
                # rc < 0 = "geometry not served" BEFORE any launch: the remembered answer was taken under other thread-local dispatch
                # options (`conv_pp_min_tiles`, `conv_pp_strips`, `conv_fwd_split_k` change what the two-group kernel takes).  Forget
                # it and run conv + head as two launches.
                self._head_ok.pop((B, Hs, Ws, hc.cout), None)
            f, _ = self.conv_bf16(t, fpn.conv[0], B, Hs, Ws, scale=sf, shift=hf, relu=True)
        L.check(lib.sd_head_fwd_bf16(f.data_ptr(), hc.weight.data_ptr(), hc.bias.data_ptr(), out.data_ptr(), B, H2 * W2, hc.cin, hc.cout,
                                     L.stream()), "sd_head_fwd_bf16")
        return out

    # ---- backward ------------------------------------------------------------------------
    def _transpose_all(self, amp):
        """Every data-gradient conv's weights [Cout][taps][Cin] -> [Cin][taps][Cout] in ONE launch at the start of a backward pass
        (42 launches of ~5 us each were launch-bound: 0.24 ms per step); _wt() then hands out views until the pass ends."""
        net = self.net
        if self._wt_plan is None:
            convs = [m for m in net.modules() if isinstance(m, ConvParams) and m.cin % 64 == 0 and m.cout % 64 == 0]
            rows, views, dst, blk = [], {}, 0, 0
            for m in convs:
                taps, n = m.k * m.k, m.weight.numel()
                nbx, nby = (m.cin + 31) // 32, (m.cout + 31) // 32
                rows.append([net._flat_off[id(m.weight)][0], dst, m.cout, taps, m.cin, blk, nbx, nby])
                views[id(m)] = (dst, n)
                dst += n; blk += nbx * nby * taps
            table = torch.tensor(rows, dtype=torch.int32, device=net.flat_params.device).contiguous()
            self._wt_plan = (table, len(convs), blk, views, dst)
        table, nconv, blocks, _, total = self._wt_plan
        key = "bf16" if amp else "f32"
        if key not in self._wt_flat:
            self._wt_flat[key] = torch.empty(total, dtype=torch.bfloat16 if amp else torch.float32, device=net.flat_params.device)
        L.check(self.lib.sd_conv2d_transpose_weights_batched(net.flat_params.data_ptr(), self._wt_flat[key].data_ptr(), table.data_ptr(), nconv, blocks,
                                                             int(amp), L.stream()), "sd_conv2d_transpose_weights_batched")
        self._wt_valid = key

    def _wt(self, conv, amp=False):
        """[Cout][taps][Cin] -> [Cin][taps][Cout] for the data-gradient (amp: written as bf16 by the same pass)."""
        if self._wt_valid == ("bf16" if amp else "f32") and id(conv) in self._wt_plan[3]:
            off, n = self._wt_plan[3][id(conv)]
            return self._wt_flat[self._wt_valid][off:off + n]
        wt = torch.empty(conv.cin * conv.k * conv.k * conv.cout, dtype=torch.bfloat16 if amp else torch.float32, device=conv.weight.device)
        transpose = self.lib.sd_conv2d_transpose_weights_bf16 if amp else self.lib.sd_conv2d_transpose_weights
        L.check(transpose(conv.weight.data_ptr(), wt.data_ptr(), conv.cout, conv.k * conv.k, conv.cin, L.stream()), "transpose_weights")
        return wt

    def _dgrad(self, dy, conv, d, res=None, bn_next=None, res_half=False):
        """dx = dgrad(dy) [+ res].  bn_next = (x, y, relu, bn, mean, invstd) of the BatchNorm whose output gradient dx is:
        the launch then also does that BatchNorm's backward reduction (sd_conv2d_dgrad_bn_reduce) and the per-channel means
        come back for _bn_bwd(..., means=...), which only has the apply pass left."""
        amp = dy.dtype == torch.bfloat16
        dx = torch.empty((d.B, d.Hi, d.Wi, conv.cin), dtype=dy.dtype, device=dy.device)
        wt = self._wt(conv, amp)
        flops = 2.0 * d.B * d.Ho * d.Wo * conv.cout * conv.cin * conv.k * conv.k
        if amp:
            assert bn_next is None, "fuse_bn_bwd is an fp32-path experiment"
            mode = 2 if res_half else (1 if res is not None else 0)
            self._timed("bf16:" + self._kname(d, 17), flops, lambda: L.check(
                self.lib.sd_conv2d_dgrad_bf16(dy.data_ptr(), wt.data_ptr(), dx.data_ptr(), C.byref(d), _ptr(res), mode, L.stream()),
                "sd_conv2d_dgrad_bf16"), phase="dgrad")
            return dx
        if bn_next is None:
            if res_half:
                fn = lambda: L.check(self.lib.sd_conv2d_dgrad_half_res(dy.data_ptr(), wt.data_ptr(), dx.data_ptr(), C.byref(d), res.data_ptr(),
                                                                       L.stream()), "sd_conv2d_dgrad_half_res")
            else:
                fn = lambda: L.check(self.lib.sd_conv2d_dgrad(dy.data_ptr(), wt.data_ptr(), dx.data_ptr(), C.byref(d), _ptr(res), L.stream()),
                                     "sd_conv2d_dgrad")
            self._timed(self._kname(d, 1), flops, fn, phase="dgrad")
            return dx
        assert not res_half
        x, y, relu, bn, mean, invstd = bn_next
        means = torch.empty(2 * conv.cin, dtype=torch.float32, device=dy.device)
        ws = self._ws(self.lib.sd_conv2d_dgrad_bn_reduce_workspace_bytes(C.byref(d)), dy.device)
        self._timed(self._kname(d, 1), flops, lambda: L.check(
            self.lib.sd_conv2d_dgrad_bn_reduce(dy.data_ptr(), wt.data_ptr(), dx.data_ptr(), C.byref(d), _ptr(res), x.data_ptr(),
                                               _ptr(y) if int(relu) in (1, 3) else 0, int(relu), mean.data_ptr(), invstd.data_ptr(),
                                               bn.weight.data_ptr(), bn.bias.data_ptr(), self.net.grad_of(bn.weight).data_ptr(),
                                               self.net.grad_of(bn.bias).data_ptr(), 0, means.data_ptr(), ws.data_ptr(), ws.numel(),
                                               L.stream()), "sd_conv2d_dgrad_bn_reduce"), phase="dgrad")
        return dx, means

    def _side_stream(self):
        if self._side is None:
            self._side = torch.cuda.Stream()
        return self._side

    def _wgrad(self, dy, x, conv, d):
        if not self.overlap_wgrad:
            return self._wgrad_now(dy, x, conv, d)
        main, side = torch.cuda.current_stream(), self._side_stream()
        ready = torch.cuda.Event()
        ready.record(main)                       # dy (and x) are complete at this point of the main stream
        with torch.cuda.stream(side):
            side.wait_event(ready)
            self._wgrad_now(dy, x, conv, d)      # L.stream() now hands the side stream to the C ABI
        dy.record_stream(side); x.record_stream(side)

    def _join_side(self):
        if self.overlap_wgrad and self._side is not None:
            torch.cuda.current_stream().wait_stream(self._side)

    def _wgrad_now(self, dy, x, conv, d):
        g = self.net.grad_of(conv.weight)
        flops = 2.0 * d.B * d.Ho * d.Wo * conv.cout * conv.cin * conv.k * conv.k
        if dy.dtype == torch.bfloat16:
            # mixed precision: bf16 operands, fp32 accumulation, fp32 gradient (3x3 / stride 1 layers on the bf16 MFMA through
            # transposed LDS reads; the strided and 1x1 convs widen their operands in the workspace and take the fp32 kernels)
            ws = self._ws(self.lib.sd_conv2d_wgrad_bf16_workspace_bytes(C.byref(d)), dy.device)
            self._timed("bf16:" + self._kname(d, 2), flops, lambda: L.check(
                self.lib.sd_conv2d_wgrad_bf16(dy.data_ptr(), x.data_ptr(), g.data_ptr(), C.byref(d), 0, ws.data_ptr(), ws.numel(), L.stream()),
                "sd_conv2d_wgrad_bf16"), phase="wgrad")
            return
        nbytes = self.lib.sd_conv2d_wgrad_workspace_bytes(C.byref(d))
        ws = self._ws(nbytes, dy.device)
        self._timed(self._kname(d, 2), flops, lambda: L.check(
            self.lib.sd_conv2d_wgrad(dy.data_ptr(), x.data_ptr(), g.data_ptr(), C.byref(d), 0, ws.data_ptr(), ws.numel(), L.stream()),
            "sd_conv2d_wgrad"), phase="wgrad")

    def _bias_grad(self, dy, conv):
        Mrows, Cc = dy.numel() // dy.shape[-1], dy.shape[-1]
        ws = self._ws(self.lib.sd_col_reduce_workspace_bytes(Mrows, Cc), dy.device)
        col_sum = self.lib.sd_col_sum_bf16 if dy.dtype == torch.bfloat16 else self.lib.sd_col_sum
        L.check(col_sum(dy.data_ptr(), Mrows, Cc, self.net.grad_of(conv.bias).data_ptr(), 0, ws.data_ptr(), ws.numel(), L.stream()),
                "sd_col_sum")

    def _bn_bwd(self, dy, x, y, relu, bn, mean, invstd, want_g=False, means=None):
        Mrows, Cc = x.numel() // x.shape[-1], x.shape[-1]
        dx = torch.empty_like(x)
        g = torch.empty_like(x) if want_g else None
        if means is not None:       # reduction (and dgamma / dbeta) already done by the data-gradient launch that produced dy
            L.check(self.lib.sd_bn_bwd_apply(dy.data_ptr(), x.data_ptr(), _ptr(y) if int(relu) in (1, 3) else 0, int(relu), Mrows, Cc,
                                             mean.data_ptr(), invstd.data_ptr(), bn.weight.data_ptr(), bn.bias.data_ptr(), means.data_ptr(),
                                             dx.data_ptr(), _ptr(g), L.stream()), "sd_bn_bwd_apply")
            return dx, g
        ws = self._ws(self.lib.sd_col_reduce_workspace_bytes(Mrows, Cc), x.device)
        # relu: False/0 = none, True/1 = mask from y, 3 = y holds the mask bytes of sd_bn_apply (residual layers), 2 = mask recomputed from x
        bn_bwd = self.lib.sd_bn_bwd_bf16 if x.dtype == torch.bfloat16 else self.lib.sd_bn_bwd
        L.check(bn_bwd(dy.data_ptr(), x.data_ptr(), _ptr(y) if int(relu) in (1, 3) else 0, int(relu), Mrows, Cc, mean.data_ptr(),
                                   invstd.data_ptr(), bn.weight.data_ptr(), bn.bias.data_ptr(), dx.data_ptr(), _ptr(g),
                                   self.net.grad_of(bn.weight).data_ptr(),
                                   self.net.grad_of(bn.bias).data_ptr(), 0, ws.data_ptr(), ws.numel(), L.stream()), "sd_bn_bwd")
        return dx, g

    def backward(self, tape, dhead, on_stage=None):
        """Writes every parameter gradient into `net.flat_grads` (overwriting, not accumulating).
        on_stage(name) is called after the gradients of a parameter group are complete, in the order
        'fpn_head', 'down4', 'down3', 'down2', 'down1_stem' (bucketed all-reduce hook)."""
        net, lib = self.net, self.lib
        B = tape["B"]
        H2, W2 = tape["hw"]
        dhead = dhead.contiguous().float()
        hc = net.head.conv
        amp = bool(tape.get("amp"))
        if amp and self.fuse_bn_bwd:
            raise L.SdError("fuse_bn_bwd is an fp32-path experiment; switch it off for mixed-precision training")
        T.push("bwd:fpn_head")
        self._transpose_all(amp)
        ws = self._ws(lib.sd_head_bwd_workspace_bytes(B, H2 * W2, hc.cin, hc.cout), dhead.device)
        if amp and hc.cin in (64, 128) and hc.cout <= 8:
            # bf16 FPN output in, bf16 gradient out (fp32 arithmetic, fp32 weight / bias gradients): no fp32 copies of the two maps
            f1 = tape["f1"]
            df = torch.empty_like(f1)
            L.check(lib.sd_head_bwd_bf16(dhead.data_ptr(), f1.data_ptr(), hc.weight.data_ptr(), df.data_ptr(), net.grad_of(hc.weight).data_ptr(),
                                         net.grad_of(hc.bias).data_ptr(), B, H2 * W2, hc.cin, hc.cout, 0, ws.data_ptr(), ws.numel(), L.stream()),
                    "sd_head_bwd_bf16")
        else:
            f1 = self._to_f32(tape["f1"]) if amp else tape["f1"]      # wide heads: fp32 kernels on widened copies
            df = torch.empty_like(f1)
            L.check(lib.sd_head_bwd(dhead.data_ptr(), f1.data_ptr(), hc.weight.data_ptr(), df.data_ptr(), net.grad_of(hc.weight).data_ptr(),
                                    net.grad_of(hc.bias).data_ptr(), B, H2 * W2, hc.cin, hc.cout, 0, ws.data_ptr(), ws.numel(), L.stream()),
                    "sd_head_bwd")
            if amp:
                df = self._to_bf16(df)

        # FPN, finest level first.  The lateral 1x1 data-gradients are deferred until the trunk's own
        # gradient for that tensor exists, so the sum of the two is the dgrad kernel's residual epilogue.
        lateral_grad = {}
        for (fpn, sc_t, (Hs, Ws), dl, t, dc, c, mf, if_, fn) in reversed(tape["fpn"]):
            dcv, _ = self._bn_bwd(df, c, fn, 2, fpn.conv[1], mf, if_)
            dt = self._dgrad(dcv, fpn.conv[0], dc)
            self._wgrad(dcv, t, fpn.conv[0], dc)
            self._wgrad(dt, sc_t, fpn.lateral, dl)
            self._bias_grad(dt, fpn.lateral)
            lateral_grad[sc_t.data_ptr()] = (dt, fpn.lateral, dl)
            dfp = torch.empty((B, Hs // 2, Ws // 2, fpn.lateral.cout), dtype=dt.dtype, device=dt.device)
            up2_bwd = lib.sd_upsample2x_bwd_bf16 if amp else lib.sd_upsample2x_bwd
            L.check(up2_bwd(dt.data_ptr(), 0, dfp.data_ptr(), B, Hs // 2, Ws // 2, fpn.lateral.cout, L.stream()), "up2_bwd")
            df = dfp
        p5, H5, W5 = tape["p5"]
        d5 = _desc(B, H5, W5, net.up1)
        self._wgrad(df, p5, net.up1, d5)
        self._bias_grad(df, net.up1)
        # With `fuse_bn_bwd` every data-gradient launch that completes the gradient of a BatchNorm output also runs that
        # BatchNorm's backward reduction in its epilogue (bn_next); `mcur` / `ma1` carry the per-channel means to the apply pass.
        blocks = tape["blocks"]

        def bn2_of(i):          # bn_next tuple of block i's second BatchNorm (mask from the saved output: residual layer)
            if i < 0 or not self.fuse_bn_bwd:
                return None
            b = blocks[i]       # (blk, xin, (Hc, Wc), d1, c1, a1, m1, i1, d2, c2, m2, i2, out, dd, cd, md, idd, relu mask bytes)
            return (b[9], b[17], 3, b[0].bn2, b[10], b[11])

        last = len(blocks) - 1
        has_lateral = blocks[last][12].data_ptr() in lateral_grad
        nxt = None if has_lateral else bn2_of(last)
        r = self._dgrad(df, net.up1, d5, bn_next=nxt)
        dcur, mcur = r if nxt is not None else (r, None)
        T.pop()
        if on_stage:
            self._join_side()
            on_stage("fpn_head")

        # trunk, last block first
        first_of = {id(net.down4[0]): "down4", id(net.down3[0]): "down3", id(net.down2[0]): "down2"}
        T.push("bwd:down4")
        for bi in range(last, -1, -1):
            (blk, xin, (Hc, Wc), d1, c1, a1, m1, i1, d2, c2, m2, i2, out, dd, cd, md, idd, msk) = blocks[bi]
            extra = lateral_grad.pop(out.data_ptr(), None)
            if extra is not None:                       # `out` also feeds an FPN lateral conv: this launch completes d(out)
                nxt = bn2_of(bi)
                r = self._dgrad(extra[0], extra[1], extra[2], res=dcur, bn_next=nxt)
                dcur, mcur = r if nxt is not None else (r, None)
            dc2, g = self._bn_bwd(dcur, c2, msk, 3, blk.bn2, m2, i2, want_g=True, means=mcur)
            nxt = (c1, None, 2, blk.bn1, m1, i1) if self.fuse_bn_bwd else None
            r = self._dgrad(dc2, blk.conv2, d2, bn_next=nxt)
            da1, ma1 = r if nxt is not None else (r, None)
            self._wgrad(dc2, a1, blk.conv2, d2)
            dc1, _ = self._bn_bwd(da1, c1, a1, 2, blk.bn1, m1, i1, means=ma1)
            half = False
            if blk.downsample is not None:
                dcd, _ = self._bn_bwd(g, cd, None, False, blk.downsample[1], md, idd)
                ds = blk.downsample[0]
                half = ds.k == 1 and ds.stride == 2 and ds.pad == 0 and Hc % 2 == 0 and Wc % 2 == 0 and not self.fuse_bn_bwd
                if half:
                    # 1x1 / stride 2: the gradient lives on the even pixels only -> stride-1 data-gradient on the small map,
                    # joined by conv1's data-gradient epilogue (no zero-filled full-size tensor)
                    dsm = L.ConvDesc()
                    dsm.B, dsm.Hi, dsm.Wi, dsm.Cin, dsm.Cout, dsm.R, dsm.S, dsm.stride, dsm.pad = B, dd.Ho, dd.Wo, ds.cin, ds.cout, 1, 1, 1, 0
                    dsm.Ho, dsm.Wo = dd.Ho, dd.Wo
                    skip = self._dgrad(dcd, ds, dsm)
                else:
                    skip = self._dgrad(dcd, ds, dd)
                self._wgrad(dcd, xin, ds, dd)
            else:
                skip = g
            # d(xin) = d(out of the previous block), complete unless that tensor also feeds an FPN lateral (handled above)
            prev_lateral = bi > 0 and blocks[bi - 1][12].data_ptr() in lateral_grad
            nxt = None if (bi == 0 or prev_lateral) else bn2_of(bi - 1)
            r = self._dgrad(dc1, blk.conv1, d1, res=skip, bn_next=nxt, res_half=half)
            dcur, mcur = r if nxt is not None else (r, None)
            self._wgrad(dc1, xin, blk.conv1, d1)
            if id(blk) in first_of:                                         # the first block of a layer closes that layer's range
                T.pop()
                if on_stage:
                    self._join_side()
                    on_stage(first_of[id(blk)])
                T.push("bwd:" + {"down4": "down3", "down3": "down2", "down2": "down1_stem"}[first_of[id(blk)]])

        # stem
        d0, s0, m0, i0, pidx = tape["stem"]
        bn0 = net.adpater[1]
        amp_stem = s0.dtype == torch.bfloat16       # bf16 conv output / pooled gradient in, bf16 gradient out (as autocast's conv1 backward sees it)
        if amp and not amp_stem:
            dcur = self._to_f32(dcur)
        ring = amp_stem and self.stem_ring and d0.Wi % 4 == 0          # the row-ring weight gradient fetches the image in aligned groups of four columns
        ds0 = torch.empty(s0.shape, dtype=torch.bfloat16 if ring else torch.float32, device=s0.device)
        ws = self._ws(lib.sd_col_reduce_workspace_bytes(B * d0.Ho * d0.Wo, 64), s0.device)
        pool_bwd = lib.sd_maxpool_bn_relu_bwd_bf16_dx16 if ring else (lib.sd_maxpool_bn_relu_bwd_bf16 if amp_stem else lib.sd_maxpool_bn_relu_bwd)
        L.check(pool_bwd(dcur.data_ptr(), pidx.data_ptr(), s0.data_ptr(), B, d0.Ho, d0.Wo, 64, m0.data_ptr(), i0.data_ptr(),
                         bn0.weight.data_ptr(), bn0.bias.data_ptr(), ds0.data_ptr(), net.grad_of(bn0.weight).data_ptr(),
                         net.grad_of(bn0.bias).data_ptr(), 0, ws.data_ptr(), ws.numel(), L.stream()), "sd_maxpool_bn_relu_bwd")
        stem = net.adpater[0]
        ws = self._ws(lib.sd_conv2d_stem_wgrad_workspace_bytes(C.byref(d0)), ds0.device)
        # amp: product on the bf16 MFMA (autocast: conv1 in bf16); a bf16 stem gives a bf16 gradient (row-ring kernel)
        stem_wgrad = lib.sd_conv2d_stem_wgrad_bf16 if ring else (lib.sd_conv2d_stem_wgrad_bf16mm if amp else lib.sd_conv2d_stem_wgrad)
        L.check(stem_wgrad(ds0.data_ptr(), tape["x"].data_ptr(), net.grad_of(stem.weight).data_ptr(), C.byref(d0), 0,
                           ws.data_ptr(), ws.numel(), L.stream()), "sd_conv2d_stem_wgrad")
        self._join_side()
        self._wt_valid = None                               # the optimizer step that follows changes the weights
        T.pop()
        if on_stage:
            on_stage("down1_stem")


class _NetFn(torch.autograd.Function):
    """One autograd node for the whole network: inputs = image + every parameter."""

    @staticmethod
    def forward(ctx, net, x, *params):
        tape = {}
        out = net._engine.forward(x, True, tape)
        ctx.net, ctx.tape = net, tape
        return out

    @staticmethod
    def backward(ctx, dhead):
        net = ctx.net
        net._engine.backward(ctx.tape, dhead)
        ctx.tape = None
        return (None, None) + tuple(net.grad_of(p) for p in net._flat_order)


class Network(nn.Module):
    def __init__(self, args, pretrained=True, raw_output: bool = False, init_weights: bool = True):
        """network.py:33.  `init_weights=False` (extension): leave the parameters uninitialised -- for callers that load a complete
        state_dict right away (`evaluate`, `detect`, `Predictor`: the seeded random init of 21.8 M values is 0.1-0.6 s of their start-up)."""
        super().__init__()
        self.raw_output = raw_output
        # bf16 backbone for the eval-mode forward (BASELINE stress config: "bf16 backbone + fp32 decode"): `--bf16_inference`, or
        # `--amp` -- the reference autocasts its validation forward under that flag too (trainer.py:141-155).  What `--amp` means
        # for the TRAINING step (trainer.py:115-121) is decided by the Trainer, not here.
        self.bf16_inference = bool(getattr(args, "use_amp", False) or getattr(args, "bf16_inference", False))
        self.label_count = len(args.labels)  # M
        self.part_count = len(args.parts)  # N
        self.out_channels = self.label_count + self.part_count + 4
        self.fpn_depth = args.fpn_depth
        if self.fpn_depth % 64:
            raise L.SdError("fpn_depth must be a multiple of 64 for the MFMA tiles")
        if self.out_channels > 32:
            raise L.SdError("labels + parts + 4 must be <= 32 (head kernel limit)")

        self.adpater = nn.Sequential(ConvParams(3, 64, 7, 2, 3), BNParams(64), Slot(), Slot())   # sic (network.py:43)
        self.down1 = _layer(64, 64, 3, 1)
        self.down2 = _layer(64, 128, 4, 2)
        self.down3 = _layer(128, 256, 6, 2)
        self.down4 = _layer(256, 512, 3, 2)
        self.up1 = ConvParams(512, self.fpn_depth, 1, bias=True)
        self.up2 = Fpn(256, self.fpn_depth)
        self.up3 = Fpn(128, self.fpn_depth)
        self.up4 = Fpn(64, self.fpn_depth)
        self.head = Head(self.fpn_depth, self.out_channels)
        if init_weights:
            self.reset_parameters(seed=0)
        # network.py:41: `resnet34(weights=ResNet34_Weights.DEFAULT if pretrained else None)` -- the ImageNet trunk.  torchvision would
        # download `resnet34-b627a593.pth` into the hub cache; here the file is looked up locally (no download): `--backbone_weights`,
        # $SDNET_BACKBONE_WEIGHTS, then torchvision's own cache locations.  Not finding it is announced loudly: the run then starts
        # from a different model than the reference's.
        self.backbone_weights = None
        if pretrained:
            path = find_backbone_weights(args)
            if path is None:
                warnings.warn(
                    "Network(pretrained=True): no ImageNet ResNet-34 checkpoint found -- the trunk starts from RANDOM weights, unlike "
                    "the reference (network.py:41).  Pass --backbone_weights /path/to/resnet34-b627a593.pth (torchvision's "
                    "ResNet34_Weights.DEFAULT state_dict), set SDNET_BACKBONE_WEIGHTS, or place the file in "
                    "$TORCH_HOME/hub/checkpoints/.", RuntimeWarning, stacklevel=2)
            else:
                self.load_backbone_state_dict(torch.load(path, map_location="cpu", weights_only=True))
                self.backbone_weights = str(path)
        self.flat_params = self.flat_grads = self.flat_params_bf16 = None
        self._flat_order, self._flat_off = [], {}
        self._folded = {}        # eval-mode (scale, shift) per BN, valid until the parameters can have changed
        self._engine = None

    # ---- init / flat storage -------------------------------------------------------------
    def reset_parameters(self, seed=0):
        """torchvision's scheme: kaiming_normal_(fan_out, relu) for convs, BN weight 1 / bias 0;
        torch default (kaiming_uniform a=sqrt(5)) for the biased FPN / head convs."""
        g = torch.Generator().manual_seed(seed)
        with torch.no_grad():
            for m in self.modules():
                if isinstance(m, ConvParams):
                    fan_out = m.cout * m.k * m.k
                    fan_in = m.cin * m.k * m.k
                    if m.bias is None:
                        m.weight.copy_(torch.randn(m.weight.shape, generator=g) * math.sqrt(2.0 / fan_out))
                    else:
                        bound = 1.0 / math.sqrt(fan_in)
                        m.weight.copy_((torch.rand(m.weight.shape, generator=g) * 2 - 1) * bound)
                        m.bias.copy_((torch.rand(m.bias.shape, generator=g) * 2 - 1) * bound)

    def _apply(self, fn, *a, **kw):
        super()._apply(fn, *a, **kw)
        self._build_flat()
        return self

    def _build_flat(self):
        """Move every parameter into one flat buffer (16-byte aligned slots) and alias it."""
        params = [p for p in self.parameters()]
        if not params or not params[0].is_cuda:
            self.flat_params = self.flat_grads = None
            return
        dev = params[0].device
        offs, total = [], 0
        for p in params:
            offs.append(total)
            total += (p.numel() + 7) // 8 * 8          # 32-byte slots: the bf16 view of a conv weight (same element offset, _w16) stays 16-byte aligned
        flat = torch.zeros(total, dtype=torch.float32, device=dev)
        grads = torch.zeros(total, dtype=torch.float32, device=dev)
        self._flat_order, self._flat_off = params, {}
        with torch.no_grad():
            for p, off in zip(params, offs):
                n = p.numel()
                if p.dim() == 4:                      # physical [Cout][R][S][Cin], logical OIHW
                    co, ci, r, s = p.shape
                    view = flat[off:off + n].view(co, r, s, ci).permute(0, 3, 1, 2)
                else:
                    view = flat[off:off + n].view(p.shape)
                view.copy_(p.data)
                p.data = view
                self._flat_off[id(p)] = (off, n)
        self.flat_params, self.flat_grads, self.flat_params_bf16 = flat, grads, None
        self._folded = {}
        self._engine = _Engine(self)

    def grad_of(self, p):
        off, n = self._flat_off[id(p)]
        g = self.flat_grads[off:off + n]
        if p.dim() == 4:
            co, ci, r, s = p.shape
            return g.view(co, r, s, ci).permute(0, 3, 1, 2)
        return g.view(p.shape)

    def invalidate_folded(self):
        """Drop the cached eval-mode BN affines (call after editing parameters in place while in eval mode)."""
        self._folded = {}

    def train(self, mode=True):
        self._folded = {}
        return super().train(mode)

    def load_state_dict(self, *a, **kw):
        self._folded = {}
        return super().load_state_dict(*a, **kw)

    def load_backbone_state_dict(self, resnet_sd):
        """Copy a torchvision ResNet-34 state_dict (keys `conv1.weight`, `bn1.*`, `layer{1..4}.{b}.*`, `fc.*`) onto the trunk, the way
        network.py:41-50 picks the sub-modules: conv1 / bn1 -> `adpater.0` / `adpater.1`, layerL -> `downL`; `fc.*` is dropped (the
        reference never registers it).  Strict on the trunk: every trunk tensor must be present with the right shape, and nothing
        but `fc.*` may be left over.  FPN and head keep their initialisation, as in the reference."""
        mapped, extra = {}, []
        for k, v in resnet_sd.items():
            if k.startswith("fc."):
                continue
            if k.startswith("conv1."):
                mapped["adpater.0." + k[len("conv1."):]] = v
            elif k.startswith("bn1."):
                mapped["adpater.1." + k[len("bn1."):]] = v
            elif k.startswith("layer") and k[5:6] in "1234" and k[6:7] == ".":
                mapped["down" + k[5:]] = v
            else:
                extra.append(k)
        own = self.state_dict()
        trunk = [k for k in own if k.split(".")[0] in ("adpater", "down1", "down2", "down3", "down4")]
        missing = [k for k in trunk if k not in mapped]
        extra += [k for k in mapped if k not in own]
        bad = [f"{k}: {tuple(mapped[k].shape)} != {tuple(own[k].shape)}" for k in trunk if k in mapped and mapped[k].shape != own[k].shape]
        if missing or extra or bad:
            raise L.SdError(f"not a torchvision ResNet-34 state_dict: missing {missing[:4]}{'...' if len(missing) > 4 else ''}, "
                            f"unexpected {extra[:4]}, shape mismatches {bad[:4]}")
        self._folded = {}
        with torch.no_grad():
            for k in trunk:
                own[k].copy_(mapped[k])
        return len(trunk)

    # ---- reference API -------------------------------------------------------------------
    def forward(self, x):  # (B, 3, H, W)
        if self._engine is None:
            raise L.SdError("Network must be moved to the GPU first (`net.to('cuda')`): there is no CPU path")
        if self.training and torch.is_grad_enabled():
            out = _NetFn.apply(self, x, *self._flat_order)
        elif self.bf16_inference and not self.training:
            out = self._engine.forward_bf16(x)
        else:
            out = self._engine.forward(x, self.training)
        if self.raw_output:
            return out
        M, nb = self.label_count, self.label_count + self.part_count
        return {"anchor_hm": out[:, :M], "part_hm": out[:, M:nb], "offsets": out[:, nb:nb + 2], "embeddings": out[:, nb + 2:nb + 4]}

    def save(self, path="last_model.pth"):
        torch.save({k: v.clone() for k, v in self.state_dict().items()}, path)

    # ---- hipGraph replay of the eval forward (launch-bound small batches) --------------------
    def graphed(self, example: torch.Tensor):
        """Capture the eval-mode forward for `example`'s shape into a hipGraph and return `run(x) -> head tensor`.
        Input and output buffers are static (the returned tensor is overwritten by the next call; write into `run.static_in` to skip the
        input copy).  Measured (tools/graph_vs_eager.sh, round 4): at bs=1 the eager stream is already gap-free -- the host enqueues ahead and
        the 44 kernels run back to back (sum of kernel durations = wall time) -- so a replay has no launch gaps to remove and adds its own
        fixed cost (the copy launch + ~9 us between replays): use it when the HOST is the bottleneck (a busy Python thread), not for speed."""
        if self.training:
            raise L.SdError("graphed() captures the inference forward: call net.eval() first")
        static_in = example.detach().clone().contiguous().float()
        # the forward `net(x)` would run: the bf16 backbone when `bf16_inference` is set (captured as it is NOW: capture again after a change)
        fwd = self._engine.forward_bf16 if self.bf16_inference else (lambda t: self._engine.forward(t, False))
        with torch.no_grad():
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):           # warm-up on the capture stream: allocations, one-time attributes
                for _ in range(2):
                    fwd(static_in)
            torch.cuda.current_stream().wait_stream(side)
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph, stream=side):
                static_out = fwd(static_in)

        def run(x):
            if x.data_ptr() != static_in.data_ptr():      # a caller that fills `run.static_in` itself saves the copy launch
                static_in.copy_(x, non_blocking=True)
            graph.replay()
            return static_out

        run.graph, run.static_in, run.static_out = graph, static_in, static_out
        return run

    # ---- explicit (autograd-free) training path used by the trainer / bench ----------------
    def forward_train(self, x, amp=False):
        """Forward in training mode recording the tape; returns (head tensor, tape).  amp=True: the mixed-precision step of the
        reference's `--amp` (trainer.py:115-121) -- bf16 activations and conv weights, fp32 accumulation, statistics and master weights."""
        tape = {}
        return self._engine.forward(x, True, tape, amp=amp), tape

    def backward_from(self, tape, dhead, on_stage=None):
        """Backward of `forward_train`: fills `flat_grads` in place."""
        self._engine.backward(tape, dhead, on_stage)

    def stage_ranges(self):
        """Flat-buffer [lo, hi) element ranges of the gradient groups reported by `backward_from(on_stage=...)`."""
        def span(mods):
            offs = [self._flat_off[id(p)] for m in mods for p in m.parameters()]
            return (min(o for o, _ in offs), max((o + n + 7) // 8 * 8 for o, n in offs))
        return {"fpn_head": span([self.up1, self.up2, self.up3, self.up4, self.head]), "down4": span([self.down4]),
                "down3": span([self.down3]), "down2": span([self.down2]), "down1_stem": span([self.adpater, self.down1])}
