#!/usr/bin/env python3
"""Headline benchmark of the SDNet hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W
    (N > 1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...)

A step = one training pass over one synthetic batch (BASELINE.json configs[2]: bs=64/GPU, 512x512 fp32,
2 labels / 1 part): HIP target rendering -> Network forward -> loss forward/backward -> Network backward
-> [RCCL all-reduce of the flat gradient buffer, 5 buckets overlapped with backward] -> fused Adam.
Inputs (images, scene keypoint arrays) are resident in HBM before the timed region.  Rank 0 prints ONE
JSON line; `value` = images/sec over all ranks (weak scaling).  The same line carries the decode
latency (us/img), the MFMA roofline of the dominant kernel measured live with stream events around
every conv launch in the timed region, and a bounded CPU baseline (the oracle, rank 0, N=1 only).
"""
import argparse
import json
import os
import sys
import time
from argparse import Namespace
from pathlib import Path

import numpy as np
import torch
import torch.distributed as dist

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))

PEAK_FP32_MFMA_TFLOPS = 157.3     # /opt/skills/guides/MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, 256 CUs x 4 SIMD x 64 FLOP/clk x 2.4 GHz
FWD_GFLOP_PER_IMG = 45.15         # conv layers only, 512x512, M+N+4 = 7
TRAIN_GFLOP_PER_IMG = 135.5


def make_args(dev, M=2, N=1, K=20, P=40):
    labels = {"bean": 0, "maize": 1} if M == 2 else {f"l{i}": i for i in range(M)}
    parts = {"leaf": 0} if N == 1 else {f"p{i}": i for i in range(N)}
    return Namespace(labels=labels, parts=parts, _r_labels={v: k for k, v in labels.items()}, _r_parts={v: k for k, v in parts.items()},
                     anchor_name="stem", down_ratio=4.0, max_objects=K, max_parts=P, conf_threshold=0.5, decoder_dist_thresh=0.1,
                     sigma_gauss=0.1, hm_loss_fn="mse", hm_weight=1.0, offset_weight=0.001, embedding_weight=0.001, fpn_depth=128,
                     learning_rate=1e-3, device=dev)


def cpu_baseline(M, N, K, P, img, budget_s=20.0):
    """Oracle (torch-CPU restatement of the reference path) timed on the host cores: train fwd+bwd on a bounded sample."""
    from oracle import sdnet_oracle as O
    # threads = the CPU share this process really has (a 1-GPU box grants 16 host cores, not the 256 it reports)
    cores = min(len(os.sched_getaffinity(0)), int(os.environ.get("SDNET_CPU_THREADS", "16")))
    torch.set_num_threads(cores)
    bs = 4
    rng = np.random.default_rng(1)
    net = O.build_reference_network(M, N).train()
    enc = O.collate([O.encode(img, img, O.synthetic_scene(rng, img, img, M, N), M, N, K, P, 4.0, 0.1) for _ in range(bs)])
    tt = {k: torch.as_tensor(v) for k, v in enc.items()}
    x = torch.randn(bs, 3, img, img)

    def one():
        for p in net.parameters():
            p.grad = None
        out = net(x)
        sig = lambda v: torch.clamp(torch.sigmoid(v), 1e-6, 1 - 1e-6)
        loss = torch.nn.functional.mse_loss(sig(out[:, :M]), tt["anchor_hm"]) + torch.nn.functional.mse_loss(sig(out[:, M:M + N]), tt["part_hm"]) \
            + 0.001 * (O._l1(out[:, M + N:M + N + 2], tt["anchor_offsets"], tt["anchor_inds"], tt["anchor_mask"])
                       + O._l1(out[:, M + N:M + N + 2], tt["part_offsets"], tt["part_inds"], tt["part_mask"])) \
            + 0.001 * O._l1(out[:, M + N + 2:], tt["embeddings"], tt["part_inds"], tt["part_mask"])
        loss.backward()

    t0 = time.perf_counter(); one(); warm = time.perf_counter() - t0
    iters = max(1, min(8, int(budget_s / max(warm, 1e-3)) - 1))
    t0 = time.perf_counter()
    for _ in range(iters):
        one()
    dt = (time.perf_counter() - t0) / iters

    # the other stages of the path: decode incl. host assembly and Encode, per image, median of 10 after 3 warm-ups
    def median_of(fn, n=10, warm=3):
        for _ in range(warm):
            fn()
        ts = []
        for _ in range(n):
            t = time.perf_counter(); fn(); ts.append(time.perf_counter() - t)
        return float(np.median(ts))

    scene = O.synthetic_scene(rng, img, img, M, N)
    one_enc = O.encode(img, img, scene, M, N, K, P, 4.0, 0.1)
    head = O.head_from_targets(rng, one_enc, M, N)[None]
    h = img // 4

    def decode_one():
        t = O.decode_tensors(head[:, :M], head[:, M:M + N], head[:, M + N:M + N + 2], head[:, M + N + 2:], K, P, 0.5, 0.1)
        return O.assemble_objects(t, 0, 0.5, 4.0, h, h)

    stages = {"decode_us_per_img": round(median_of(decode_one) * 1e6, 1),
              "encode_us_per_img": round(median_of(lambda: O.encode(img, img, scene, M, N, K, P, 4.0, 0.1)) * 1e6, 1)}
    return {"value": round(bs / dt, 3), "unit": "images/sec", "cores": cores, "kind": "port",
            "sample": f"oracle ReferenceNetwork + loss, train fwd+bwd, bs={bs} {img}x{img} fp32, {iters} timed iters after 1 warm-up, torch {torch.__version__} CPU",
            "stages": stages}


PMC_TRAFFIC = Path(__file__).resolve().parent / "profiles" / "r01_pmc_hbm_traffic_conv_kernels.json"


def pmc_traffic(kind):
    """HBM bytes per launch of the dominant kernel kind (e.g. "k_conv_igemm<128>": forward + data-gradient instantiations,
    launch-weighted) from the committed rocprofv3 PMC passes of this same workload (tools/pmc_traffic.sh: FETCH_SIZE x 2 for
    gfx950 + WRITE_SIZE, one counter per pass).  PMC counters cannot be read from inside the process, so bench.py reports the
    profile's figure; None when the profile is absent."""
    try:
        prof = json.loads(PMC_TRAFFIC.read_text())
    except OSError:
        return None, None
    stem = "sd::" + kind.rstrip(">")
    rows = [v for k, v in prof.items() if k.startswith(stem + ",") or k.startswith("sd::" + kind + "<") or k == "sd::" + kind]
    n = sum(v["launches"] for v in rows)
    if not n:
        return None, None
    return round(sum(v["launches"] * v["hbm_bytes_per_launch"] for v in rows) / n), "profiles/" + PMC_TRAFFIC.name


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=64, help="images per GPU")
    ap.add_argument("--size", type=int, default=512)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--overlap-wgrad", action="store_true",
                    help="run the weight-gradient GEMMs on a side stream (+3.5 %% images/s; per-kernel durations then overlap)")
    ap.add_argument("--exchange", choices=("torch", "rccl"), default=None,
                    help="gradient all-reduce binding for --gpus > 1: torch.distributed's RCCL (default) or the C-ABI sd_allreduce_*")
    ap.add_argument("--fuse-bn-bwd", dest="fuse_bn_bwd", action="store_true", default=None,
                    help="BatchNorm-backward reductions inside the data-gradient epilogues (experiment switch; default: the engine's)")
    ap.add_argument("--no-fuse-bn-bwd", dest="fuse_bn_bwd", action="store_false")
    ap.add_argument("--zero-input", action="store_true",
                    help="experiment: all-zero images (every activation is then zero): same kernels at lower MFMA power -> DVFS headroom")
    ap.add_argument("--extras", action="store_true",
                    help="also time the eval-mode forward (bs=B and bs=1) after the timed region; off by default so that the "
                         "rocprofv3 per-kernel averages of the default command describe the timed training steps only")
    a = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X visible as torch device 'cuda' (there is no CPU path to measure)")
    dev = torch.device("cuda", local if local < torch.cuda.device_count() else 0)    # (rehearsal: several ranks on one GPU)
    torch.cuda.set_device(dev)
    if world > 1:
        backend = os.environ.get("SDNET_DIST_BACKEND", "nccl")      # "gloo": rehearsal of the multi-rank path on one GPU / on CPU hosts
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)
    assert a.gpus == world, f"--gpus {a.gpus} but WORLD_SIZE={world}"

    from structuredetector_amd.data import Decoder, Encode
    from structuredetector_amd.data.synthetic import synthetic_batch
    from structuredetector_amd.model import Network
    from structuredetector_amd.model.trainer import TrainStep

    M, N, K, P, B, img = 2, 1, 20, 40, a.batch, a.size
    args = make_args(dev, M, N, K, P)
    torch.manual_seed(926354916)                         # args.py:257; identical init on every rank (+ broadcast)
    net = Network(args, pretrained=False).to(dev).train()
    step = TrainStep(net, args, exchange=a.exchange)
    step.sync_parameters()
    net._engine.overlap_wgrad = bool(a.overlap_wgrad)
    if a.fuse_bn_bwd is not None:
        net._engine.fuse_bn_bwd = bool(a.fuse_bn_bwd)
    enc = Encode(args)
    rng = np.random.default_rng(926354916 + rank)        # per-rank data
    gen = torch.Generator(device=dev).manual_seed(926354916 + rank)
    images = torch.randn(B, 3, img, img, device=dev, generator=gen)
    if a.zero_input:
        images.zero_()
    plans = [enc.upload(enc.plan(img, img, *synthetic_batch(rng, B, img, img, M, N))) for _ in range(4)]
    torch.cuda.synchronize()

    def run_step(i):
        return step(images, enc.render_device(plans[i % len(plans)]))

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for i in range(a.warmup):
        loss = run_step(i)
    barrier()
    net._engine.prof = []                                # stream events around every conv launch of the timed region
    t0 = time.perf_counter()
    for i in range(a.steps):
        loss = run_step(a.warmup + i)
    barrier()
    dt = time.perf_counter() - t0
    prof, net._engine.prof = net._engine.prof, None
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    loss_host = [float(v) for v in loss.cpu()]
    assert all(np.isfinite(loss_host)), f"non-finite loss {loss_host}"

    # ---- roofline of the dominant kernel (live, from the timed region)
    per, phases = {}, {}
    for kind, flops, e0, e1, phase in prof:
        t = e0.elapsed_time(e1) * 1e-3
        d = per.setdefault(kind, [0, 0.0, 0.0])
        d[0] += 1; d[1] += flops; d[2] += t
        q = phases.setdefault(phase, [0.0, 0.0])
        q[0] += flops; q[1] += t
    kernels = {k: {"launches_per_step": v[0] // max(a.steps, 1), "avg_launch_us": round(v[2] / v[0] * 1e6, 2),
                   "gflop_per_launch": round(v[1] / v[0] / 1e9, 3), "tflops": round(v[1] / v[2] / 1e12, 2)} for k, v in per.items()}
    dom = max(per, key=lambda k: per[k][2])
    ach = per[dom][1] / per[dom][2] / 1e12
    conv_time_frac = sum(v[2] for v in per.values()) / (dt if world == 1 else max(dt, 1e-9))
    traffic, traffic_src = pmc_traffic(dom) if (B, img) == (64, 512) else (None, None)
    roofline = {"bound": "mfma", "kernel": dom, "achieved": round(ach, 2), "peak": PEAK_FP32_MFMA_TFLOPS, "unit": "TFLOP/s",
                "frac": round(ach / PEAK_FP32_MFMA_TFLOPS, 4), "traffic": traffic, "traffic_source": traffic_src,
                "flops_per_launch": round(per[dom][1] / per[dom][0], 1), "avg_launch_us": kernels[dom]["avg_launch_us"],
                "launches_per_step": kernels[dom]["launches_per_step"], "all_conv_kernels": kernels,
                "conv_phases": {ph: {"ms_per_step": round(v[1] / a.steps * 1e3, 3), "tflops": round(v[0] / v[1] / 1e12, 2),
                                     "frac_of_peak": round(v[0] / v[1] / 1e12 / PEAK_FP32_MFMA_TFLOPS, 4)} for ph, v in phases.items()},
                "conv_share_of_step_time": round(conv_time_frac, 3)}

    # ---- forward-only and decode figures (same process, after the timed region)
    extra = {}
    if rank == 0 and a.extras:
        net.eval()
        with torch.no_grad():
            for _ in range(2):
                out = net(images)
            torch.cuda.synchronize(); t1 = time.perf_counter()
            for _ in range(3):
                out = net(images)
            torch.cuda.synchronize()
            fwd = (time.perf_counter() - t1) / 3
        extra["fwd_eval_ms_bs%d" % B] = round(fwd * 1e3, 3)
        extra["fwd_eval_tflops"] = round(B * FWD_GFLOP_PER_IMG / fwd / 1e3, 2)
        extra["fwd_eval_frac_of_mfma_peak"] = round(B * FWD_GFLOP_PER_IMG / fwd / 1e3 / PEAK_FP32_MFMA_TFLOPS, 4)
        # BASELINE configs[1]: bs=1 inference latency (split-K convs fill the chip at small M)
        dec1 = Decoder(args)
        with torch.no_grad():
            x1 = images[:1].contiguous()
            for _ in range(3):
                net(x1)
            torch.cuda.synchronize(); t1 = time.perf_counter()
            for _ in range(10):
                o1 = net(x1)
            torch.cuda.synchronize()
            extra["fwd_eval_ms_bs1"] = round((time.perf_counter() - t1) / 10 * 1e3, 3)
            t1 = time.perf_counter()
            for _ in range(10):
                dec1(net(x1))
            extra["infer_e2e_ms_per_img_bs1"] = round((time.perf_counter() - t1) / 10 * 1e3, 3)     # forward + decode + host objects
        net.train()
    if rank == 0:
        dec = Decoder(args)
        tgt = enc.render_device(plans[0])
        hm = torch.cat([tgt["anchor_hm"], tgt["part_hm"]], 1).clamp(1e-4, 0.95)
        head = torch.cat([torch.log(hm / (1 - hm)) + 0.05 * torch.randn(hm.shape, device=dev, generator=gen),
                          0.1 * torch.randn(B, 4, img // 4, img // 4, device=dev, generator=gen)], 1)
        outs = {"anchor_hm": head[:, :M], "part_hm": head[:, M:M + N], "offsets": head[:, M + N:M + N + 2], "embeddings": head[:, M + N + 2:]}
        for _ in range(3):
            dec.decode_packed(outs, 0.5, 0.1, exact_topk=False)      # what Decoder.__call__ runs when no metadata is requested
        torch.cuda.synchronize(); t1 = time.perf_counter()
        for _ in range(20):
            dec.decode_packed(outs, 0.5, 0.1, exact_topk=False)
        torch.cuda.synchronize()
        d_dev = (time.perf_counter() - t1) / 20
        extra["decode_device_us_per_img_bs%d" % B] = round(d_dev / B * 1e6, 3)
        extra["decode_device_GBps_bs%d" % B] = round(B * 199008 / d_dev / 1e9, 1)       # 199,008 B/img
        one = {k: v[:1] for k, v in outs.items()}
        for _ in range(3):
            dec(one)
        t1 = time.perf_counter()
        for _ in range(20):
            dec(one)
        extra["decode_e2e_us_per_img_bs1"] = round((time.perf_counter() - t1) / 20 * 1e6, 1)   # 2 launches + D2H + host assembly

    if rank == 0:
        imgs_per_s = B * world * a.steps / dt
        line = {
            "metric": "images/sec (train fwd+bwd) and decode us/img at 512x512", "value": round(imgs_per_s, 2), "unit": "images/sec",
            "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": round(dt / a.steps * 1e3, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"configs[2]: train step bs={B}/GPU {img}x{img} fp32, 2 labels / 1 part, K=20 P=40, "
                                   "render targets + fwd + MSE/L1 loss + bwd + Adam; random-init ResNet-34+FPN",
                       "global_batch": B * world, "parallelism": f"dp{world}", "overlap_wgrad": bool(a.overlap_wgrad), "fuse_bn_bwd": bool(net._engine.fuse_bn_bwd), **({"zero_input": True} if a.zero_input else {}),
                       "exchange": "sd_allreduce (RCCL via C ABI)" if step.rccl is not None else (f"torch.distributed {dist.get_backend()}" if world > 1 else "none")},
            "train_tflops_per_gpu": round(B * TRAIN_GFLOP_PER_IMG * a.steps / dt / 1e3, 2),
            "train_frac_of_mfma_peak": round(B * TRAIN_GFLOP_PER_IMG * a.steps / dt / 1e3 / PEAK_FP32_MFMA_TFLOPS, 4),
            "loss": [round(v, 6) for v in loss_host],
            "roofline": roofline,
        }
        line.update(extra)
        if world == 1 and not a.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(M, N, K, P, img)
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
