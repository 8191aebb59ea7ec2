#!/usr/bin/env python3
"""Headline benchmark of the SDNet hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W
    (N > 1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...)

A step = one training pass over one synthetic batch (BASELINE.json configs[2]: bs=64/GPU, 512x512 fp32,
2 labels / 1 part): HIP target rendering -> Network forward -> loss forward/backward -> Network backward
-> [RCCL all-reduce of the flat gradient buffer, 5 buckets overlapped with backward] -> fused Adam.
Inputs (images, scene keypoint arrays) are resident in HBM before the timed region.  Rank 0 prints ONE
JSON line; `value` = images/sec over all ranks (weak scaling).  The same line carries
  * `roofline`: the MFMA roofline of the dominant kernel, measured live with stream events around every conv launch
    of the timed region (+ every conv kernel and the fwd / dgrad / wgrad phases);
  * `rccl`: self-evidence of the gradient exchange -- before the timed region every rank puts rank+1 through the
    SAME buckets / binding / streams the step uses and the sum must be N(N+1)/2 on every rank; after it, the isolated
    duration of each bucket's all-reduce and `exposed_comm_ms` = step time with the exchange - step time without;
  * `north_star` (N=1, after the timed region; `--no-extras` skips it): eval forward bs=64 fp32 vs the fp32 MFMA peak
    (the north-star's >= 60 % target), bs=1 forward + decode + objects (configs[1]), bf16 forward bs=64 vs the bf16
    MFMA peak, and configs[4] (1024x1024, 8 labels / 8 parts, K=128, P=512, bs=16: bf16 forward + fp32 decode);
  * `decode`: HIP decoder us/img (bs=64 on the device, bs=1 end to end) and its fraction of the HBM roofline;
  * `cpu_baseline`: the oracle timed on the host cores, SURVEY.md 8(d) protocol (rank 0, N=1 only).
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
PEAK_BF16_MFMA_TFLOPS = 2500.0    # same guide: dense bf16 MFMA (the headline 5 PFLOP/s figure includes 2:1 sparsity)
PEAK_HBM_GBPS = 8000.0            # HBM3E spec
FWD_GFLOP_PER_IMG = 45.15         # conv layers only, 512x512, M+N+4 = 7
TRAIN_GFLOP_PER_IMG = 135.5
STRESS_FWD_GFLOP_PER_IMG = 180.8  # 1024x1024, 8 labels / 8 parts
DECODE_BYTES_PER_IMG = 199008     # heatmap logits read once + gathers + outputs (K=20, P=40)
STRESS_DECODE_BYTES_PER_IMG = 4221952


def make_args(dev, M=2, N=1, K=20, P=40):
    labels = {"bean": 0, "maize": 1} if M == 2 else {f"l{i}": i for i in range(M)}
    parts = {"leaf": 0} if N == 1 else {f"p{i}": i for i in range(N)}
    return Namespace(labels=labels, parts=parts, _r_labels={v: k for k, v in labels.items()}, _r_parts={v: k for k, v in parts.items()},
                     anchor_name="stem", down_ratio=4.0, max_objects=K, max_parts=P, conf_threshold=0.5, decoder_dist_thresh=0.1,
                     sigma_gauss=0.1, hm_loss_fn="mse", hm_weight=1.0, offset_weight=0.001, embedding_weight=0.001, fpn_depth=128,
                     learning_rate=1e-3, device=dev)


def cpu_baseline(M, N, K, P, img, budget_s=24.0):
    """Oracle (torch-CPU restatement of the reference path) timed on the host cores, SURVEY.md 8(d) protocol: stages timed
    separately, 3 warm-ups + the median of 10 iterations each (fewer only when one iteration alone would break the time
    budget -- the counts actually used are reported), all granted cores plus a 1-thread figure.  `value` = train fwd+bwd
    images/sec (network + loss, bs=8).  A reported baseline, not the optimisation target."""
    from oracle import sdnet_oracle as O
    # threads = the CPU share this process really has (a 1-GPU box grants 16 host cores, not the 256 it reports)
    cores = min(len(os.sched_getaffinity(0)), int(os.environ.get("SDNET_CPU_THREADS", "16")))
    torch.set_num_threads(cores)
    t_start = time.perf_counter()
    rng = np.random.default_rng(1)
    net = O.build_reference_network(M, N).train()

    def median_of(fn, n=10, warm=3, cap_s=None):
        """3 warm-ups + median of n; when cap_s is given and the warm-ups show that n iterations would exceed it, n shrinks (>= 3)."""
        t0 = time.perf_counter()
        for _ in range(warm):
            fn()
        per = (time.perf_counter() - t0) / max(warm, 1)
        if cap_s is not None:
            n = int(max(3, min(n, cap_s / max(per, 1e-6))))
        ts = []
        for _ in range(n):
            t = time.perf_counter(); fn(); ts.append(time.perf_counter() - t)
        return float(np.median(ts)), n

    def batch(bs):
        enc = O.collate([O.encode(img, img, O.synthetic_scene(rng, img, img, M, N), M, N, K, P, 4.0, 0.1) for _ in range(bs)])
        return torch.randn(bs, 3, img, img), {k: torch.as_tensor(v) for k, v in enc.items()}

    sig = lambda v: torch.clamp(torch.sigmoid(v), 1e-6, 1 - 1e-6)

    def loss_of(out, tt):
        return torch.nn.functional.mse_loss(sig(out[:, :M]), tt["anchor_hm"]) + torch.nn.functional.mse_loss(sig(out[:, M:M + N]), tt["part_hm"]) \
            + 0.001 * (O._l1(out[:, M + N:M + N + 2], tt["anchor_offsets"], tt["anchor_inds"], tt["anchor_mask"])
                       + O._l1(out[:, M + N:M + N + 2], tt["part_offsets"], tt["part_inds"], tt["part_mask"])) \
            + 0.001 * O._l1(out[:, M + N + 2:], tt["embeddings"], tt["part_inds"], tt["part_mask"])

    x8, t8 = batch(8)
    x1 = x8[:1].contiguous()

    def train8():
        for p in net.parameters():
            p.grad = None
        loss_of(net(x8), t8).backward()

    stages, counts = {}, {}
    dt_train, counts["train_fwd_bwd_bs8"] = median_of(train8, cap_s=0.45 * budget_s)
    stages["train_fwd_bwd_ms_bs8"] = round(dt_train * 1e3, 1)
    net.eval()
    with torch.no_grad():
        d, counts["backbone_fwd_bs8"] = median_of(lambda: net(x8), cap_s=0.15 * budget_s)
        stages["backbone_fwd_ms_bs8"] = round(d * 1e3, 1)
        d, counts["backbone_fwd_bs1"] = median_of(lambda: net(x1), cap_s=0.08 * budget_s)
        stages["backbone_fwd_ms_bs1"] = round(d * 1e3, 1)
        head8 = net(x8)
    # Loss forward + backward alone (loss.py:17-50), bs=8, on a head tensor of the real shape
    hl = head8.detach().clone().requires_grad_(True)

    def loss8():
        hl.grad = None
        loss_of(hl, t8).backward()

    d, counts["loss_fwd_bwd_bs8"] = median_of(loss8)
    stages["loss_fwd_bwd_ms_bs8"] = round(d * 1e3, 2)
    scene = O.synthetic_scene(rng, img, img, M, N)
    one_enc = O.encode(img, img, scene, M, N, K, P, 4.0, 0.1)
    head = O.head_from_targets(rng, one_enc, M, N)[None]
    h = img // 4

    def decode_one():
        t = O.decode_tensors(head[:, :M], head[:, M:M + N], head[:, M + N:M + N + 2], head[:, M + N + 2:], K, P, 0.5, 0.1)
        return O.assemble_objects(t, 0, 0.5, 4.0, h, h)

    d, counts["decode"] = median_of(decode_one)
    stages["decode_us_per_img"] = round(d * 1e6, 1)
    d, counts["encode"] = median_of(lambda: O.encode(img, img, scene, M, N, K, P, 4.0, 0.1))
    stages["encode_us_per_img"] = round(d * 1e6, 1)
    # 1-thread figures: train fwd+bwd at bs=1 and the eval forward at bs=1
    torch.set_num_threads(1)
    net.train()
    xt, tt1 = x8[:1].contiguous(), {k: v[:1] for k, v in t8.items()}

    def train1():
        for p in net.parameters():
            p.grad = None
        loss_of(net(xt), tt1).backward()

    d, counts["train_fwd_bwd_bs1_1thread"] = median_of(train1, n=5, warm=1, cap_s=0.15 * budget_s)
    stages["train_fwd_bwd_ms_bs1_1thread"] = round(d * 1e3, 1)
    one_thread = 1.0 / d
    net.eval()
    with torch.no_grad():
        d, counts["backbone_fwd_bs1_1thread"] = median_of(lambda: net(x1), n=5, warm=1, cap_s=0.06 * budget_s)
    stages["backbone_fwd_ms_bs1_1thread"] = round(d * 1e3, 1)
    torch.set_num_threads(cores)
    return {"value": round(8 / dt_train, 3), "unit": "images/sec", "cores": cores, "kind": "port",
            "value_1thread": round(one_thread, 3),
            "sample": f"oracle ReferenceNetwork + loss, train fwd+bwd, bs=8 {img}x{img} fp32, 3 warm-ups + median of "
                      f"{counts['train_fwd_bwd_bs8']} iterations on {cores} threads (1-thread figure: bs=1); stages timed separately; "
                      f"torch {torch.__version__} CPU; {round(time.perf_counter() - t_start, 1)} s of CPU work",
            "stages": stages, "iterations": counts,
            "stages_are": "timings of THIS repository's CPU oracle (oracle/sdnet_oracle.py: numpy index work, torch-CPU fp32 maps), not of the "
                          "reference's torch + .item() code (BASELINE.md: 5.8 ms decode / 11.4 ms encode per image on another box)"}


PROFILES = Path(__file__).resolve().parent / "profiles"
CONV_SOURCE = Path(__file__).resolve().parent / "structuredetector_amd" / "csrc" / "sd_conv.hip"


def pmc_traffic(kind):
    """HBM bytes per launch of the dominant kernel (exact rocprofv3 kernel name, e.g. "k_conv3x3_patch<128, false>") from the
    newest committed PMC profile of this same workload (tools/pmc_traffic.sh: FETCH_SIZE x 2 for gfx950 + WRITE_SIZE, one
    counter per pass).  PMC counters cannot be read from inside the process, so bench.py reports the profile's figure -- but
    only while it still describes the code: the profile records the sha256 of csrc/sd_conv.hip it was taken with, and a
    profile of a different source (or without the kernel) yields traffic = null plus the reason."""
    import hashlib
    cands = sorted(PROFILES.glob("r*_pmc_hbm_traffic_conv_kernels.json"), reverse=True)
    if not cands:
        return None, "no PMC profile committed"
    prof = json.loads(cands[0].read_text())
    src = "profiles/" + cands[0].name
    want = prof.get("_meta", {}).get("sd_conv_hip_sha256")
    have = hashlib.sha256(CONV_SOURCE.read_bytes()).hexdigest()
    if want != have:
        return None, f"{src} is stale: taken with another csrc/sd_conv.hip (sha256 {str(want)[:12]} != {have[:12]})"
    row = prof.get("sd::" + kind)
    if row is None:
        return None, f"{src} has no row for {kind}"
    return round(row["hbm_bytes_per_launch"]), src


def self_launch(n, argv):
    """Spawn `python -m torch.distributed.run --nnodes=1 --nproc-per-node n ... bench.py <argv>` (the driver's own launch line),
    wait, and return the child's exit code.  No torch.cuda call happens in this (parent) process."""
    import socket
    import subprocess
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), str(Path(__file__).resolve()), *argv]
    print(f"bench.py: --gpus {n} without a launcher: spawning {' '.join(cmd[1:7])} ...", file=sys.stderr, flush=True)
    return subprocess.run(cmd, env=env).returncode


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=64, help="images per GPU")
    ap.add_argument("--size", type=int, default=512)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--overlap-wgrad", action="store_true",
                    help="experiment: run the weight-gradient GEMMs on a side stream (measured in round 2: fp32 step 6 %% SLOWER, mixed "
                         "precision -0.3 %%: both streams are MFMA-bound and share the power budget; per-kernel durations then overlap)")
    ap.add_argument("--exchange", choices=("torch", "rccl"), default=None,
                    help="gradient all-reduce binding for --gpus > 1: torch.distributed's RCCL (default) or the C-ABI sd_allreduce_*")
    ap.add_argument("--no-comm-sim", dest="comm_sim", action="store_false", default=True,
                    help="skip `north_star.comm_sim` (single-GPU runs): the training step with a simulated all-reduce beside its backward")
    ap.add_argument("--comm-sim-wgs", type=int, default=32, help="workgroups of the simulated collective (RCCL: one or two per channel)")
    ap.add_argument("--comm-sim-gbps", type=float, default=200.0,
                    help="modelled bus bandwidth of the 8-rank all-reduce over xGMI (7 links x ~50 GB/s per direction: 150-300 GB/s for 1-50 MB buckets)")
    ap.add_argument("--fuse-bn-bwd", dest="fuse_bn_bwd", action="store_true", default=None,
                    help="BatchNorm-backward reductions inside the data-gradient epilogues (experiment switch; default: the engine's)")
    ap.add_argument("--no-fuse-bn-bwd", dest="fuse_bn_bwd", action="store_false")
    ap.add_argument("--amp", action="store_true",
                    help="time the mixed-precision training step instead (reference `--amp`: bf16 activations / conv weights, fp32 "
                         "accumulation and master weights); the default line is the fp32 step BASELINE.json names")
    ap.add_argument("--zero-input", action="store_true",
                    help="experiment: all-zero images (every activation is then zero): same kernels at lower MFMA power -> DVFS headroom")
    ap.add_argument("--no-conv-events", action="store_true",
                    help="A/B of the measurement itself: no stream events around the conv launches of the timed region (the per-kernel "
                         "figures then come from three extra steps; DESIGN.md records the difference: events on every step cost 1 %%)")
    ap.add_argument("--no-extras", dest="extras", action="store_false", default=True,
                    help="skip the north-star side figures measured AFTER the timed region (eval forward bs=B / bs=1, bf16 forward, "
                         "configs[4] stress forward + decode): use it under rocprofv3 --stats so that the per-kernel averages "
                         "describe the timed training steps only")
    a = ap.parse_args()

    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # `python bench.py --gpus N` without a launcher: start the N ranks ourselves, as a CHILD `torch.distributed.run`, BEFORE
        # this process makes any GPU call (a process that has initialised HIP must never be replaced by another program; a
        # parent that has not touched the card may simply wait).  The child's rank 0 prints the one JSON line on our stdout.
        raise SystemExit(self_launch(a.gpus, sys.argv[1:]))

    # A rank that hangs (a collective that never completes, a peer that died) would otherwise sit silently until the launcher's limit:
    # after SDNET_BENCH_WATCHDOG seconds (default 20 minutes; 0 = off) every thread's Python stack goes to stderr and the process exits.
    import faulthandler
    watchdog = float(os.environ.get("SDNET_BENCH_WATCHDOG", "1200"))
    if watchdog > 0:
        faulthandler.dump_traceback_later(watchdog, exit=True)

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if a.gpus != world:                                   # checked before the first GPU call
        raise SystemExit(f"bench.py: --gpus {a.gpus} but the launcher set WORLD_SIZE={world}")
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X visible as torch device 'cuda' (there is no CPU path to measure)")
    backend = os.environ.get("SDNET_DIST_BACKEND", "nccl")          # "gloo": rehearsal of the multi-rank path on one GPU / on CPU hosts
    if local >= torch.cuda.device_count():
        # RCCL needs one device per rank: a rank silently sharing device 0 would time a different experiment
        if world > 1 and backend == "nccl":
            raise SystemExit(f"bench.py: LOCAL_RANK {local} but only {torch.cuda.device_count()} visible device(s); the RCCL run needs one "
                             "GPU per rank (set SDNET_DIST_BACKEND=gloo to rehearse several ranks on one GPU)")
        local = 0                                                    # (gloo rehearsal: several ranks on one GPU)
    dev = torch.device("cuda", local)
    torch.cuda.set_device(dev)
    if world > 1:
        # eight ranks share the host: cap each rank's CPU pools (the Encode planner and torch's intra-op pool) at its share of the cores
        share = max(1, (os.cpu_count() or 8) // world)
        torch.set_num_threads(min(torch.get_num_threads(), share))
        os.environ.setdefault("OMP_NUM_THREADS", str(share))
    if world > 1:
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)

    from structuredetector_amd.data import Decoder, Encode
    from structuredetector_amd.data.synthetic import synthetic_batch
    from structuredetector_amd.model import Network
    from structuredetector_amd.model.trainer import TrainStep

    M, N, K, P, B, img = 2, 1, 20, 40, a.batch, a.size
    args = make_args(dev, M, N, K, P)
    args.use_amp = bool(a.amp)
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

    # ---- self-evidence of the exchange that is about to be timed: sum(rank+1) through the SAME buckets / binding / streams
    rccl = step.verify_exchange()
    if world > 1:
        backend = dist.get_backend()
        rccl["backend"] = backend
        if backend == "nccl":
            try:
                rccl["version"] = ".".join(str(v) for v in torch.cuda.nccl.version())  # torch's "nccl" IS RCCL on ROCm
            except Exception as err:
                rccl["version"] = f"unavailable ({err!r})"
        devs = torch.tensor([torch.cuda.current_device(), torch.cuda.device_count()], device=dev)
        gathered = [torch.zeros_like(devs) for _ in range(world)]
        dist.all_gather(gathered, devs)
        rccl["device_of_rank"] = [int(g[0]) for g in gathered]
        rccl["distinct_devices"] = len({int(g[0]) for g in gathered})
        if backend == "nccl" and rccl["distinct_devices"] != world:
            raise SystemExit(f"bench.py: {world} ranks over RCCL but {rccl['distinct_devices']} distinct devices ({rccl['device_of_rank']})")

    for i in range(a.warmup):
        loss = run_step(i)
    barrier()
    # Stream events around every conv launch of the timed region, on every FOURTH step: two hipEventRecords around each of the
    # ~127 conv launches cost 0.8 ms per step when every step carries them (same-box A/B, `--no-conv-events`: 844 vs 853 img/s), i.e.
    # the measurement lowered the number it reports by 1 %; sampled, it costs 0.25 %.  The per-kernel figures are averages over the
    # sampled steps' launches (still launches OF the timed region, on the stream they run on).
    events = []
    every = 1 if a.steps < 8 else 4
    sampled = 0
    t0 = time.perf_counter()
    for i in range(a.steps):
        on = (not a.no_conv_events) and i % every == 0
        net._engine.prof = events if on else None
        sampled += int(on)
        loss = run_step(a.warmup + i)
    net._engine.prof = None
    barrier()
    dt = time.perf_counter() - t0
    prof, prof_steps = events, sampled
    if not prof:                                         # --no-conv-events: the per-kernel figures come from three extra (untimed) steps
        net._engine.prof = prof = []
        prof_steps = 3
        for i in range(prof_steps):
            run_step(i)
        barrier()
        net._engine.prof = None
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        per_rank = [torch.zeros_like(t) for _ in range(world)]
        dist.all_gather(per_rank, t)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
        rccl["ms_per_step_per_rank_min"] = round(min(float(v.item()) for v in per_rank) / a.steps * 1e3, 3)
        rccl["ms_per_step_per_rank_max"] = round(max(float(v.item()) for v in per_rank) / a.steps * 1e3, 3)
    loss_host = [float(v) for v in loss.cpu()]
    assert all(np.isfinite(loss_host)), f"non-finite loss {loss_host}"
    if world > 1:
        # exposed communication = step time with the exchange - step time without it (same schedule, all-reduce skipped),
        # and the isolated duration of each bucket's all-reduce (what the overlap has to hide)
        rccl["buckets_isolated"] = step.time_buckets()
        step.exchange_enabled = False
        k2 = max(3, min(a.steps, 10))
        run_step(0)
        barrier()
        t1 = time.perf_counter()
        for i in range(k2):
            run_step(i)
        barrier()
        t_no = torch.tensor([(time.perf_counter() - t1) / k2], dtype=torch.float64, device=dev)
        dist.all_reduce(t_no, op=dist.ReduceOp.MAX)
        step.exchange_enabled = True
        rccl["ms_per_step_without_exchange"] = round(float(t_no.item()) * 1e3, 3)
        rccl["exposed_comm_ms"] = round(dt / a.steps * 1e3 - float(t_no.item()) * 1e3, 3)
        rccl["allreduce_bytes_per_step"] = int(net.flat_grads.numel()) * 4

    # ---- roofline of the dominant kernel (live, from the timed region)
    per, phases = {}, {}
    for kind, flops, e0, e1, phase in prof:
        t = e0.elapsed_time(e1) * 1e-3
        d = per.setdefault(kind, [0, 0.0, 0.0])
        d[0] += 1; d[1] += flops; d[2] += t
        q = phases.setdefault(phase, [0.0, 0.0])
        q[0] += flops; q[1] += t
    kernels = {k: {"launches_per_step": v[0] // max(prof_steps, 1), "avg_launch_us": round(v[2] / v[0] * 1e6, 2),
                   "gflop_per_launch": round(v[1] / v[0] / 1e9, 3), "tflops": round(v[1] / v[2] / 1e12, 2)} for k, v in per.items()}
    dom = max(per, key=lambda k: per[k][2])
    ach = per[dom][1] / per[dom][2] / 1e12
    # (`--amp`: the engine labels its bf16 launches "bf16:<kernel the fp32 dispatcher would pick>"; their roofline is the bf16 MFMA peak)
    peak_dom = PEAK_BF16_MFMA_TFLOPS if dom.startswith("bf16:") else PEAK_FP32_MFMA_TFLOPS
    conv_time_frac = sum(v[2] for v in per.values()) / prof_steps / max(dt / a.steps, 1e-9)
    traffic, traffic_src = pmc_traffic(dom) if (B, img) == (64, 512) and not a.amp else (None, None)
    roofline = {"bound": "mfma", "kernel": dom, "achieved": round(ach, 2), "peak": peak_dom, "unit": "TFLOP/s",
                "frac": round(ach / peak_dom, 4), "traffic": traffic, "traffic_source": traffic_src,
                "flops_per_launch": round(per[dom][1] / per[dom][0], 1), "avg_launch_us": kernels[dom]["avg_launch_us"],
                "launches_per_step": kernels[dom]["launches_per_step"], "all_conv_kernels": kernels,
                "conv_phases": {ph: {"ms_per_step": round(v[1] / prof_steps * 1e3, 3), "tflops": round(v[0] / v[1] / 1e12, 2),
                                     "frac_of_peak": round(v[0] / v[1] / 1e12 / peak_dom, 4)} for ph, v in phases.items()},
                "conv_share_of_step_time": round(conv_time_frac, 3)}

    # ---- north-star side figures (same process, AFTER the timed region; rank 0, single-GPU runs only)
    extra = {}

    def timed(fn, n, warm=2):
        for _ in range(warm):
            fn()
        torch.cuda.synchronize(); t1 = time.perf_counter()
        for _ in range(n):
            fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t1) / n

    _blk = {}

    def timed_device(fn, n, warm=2):
        """Device time per call of a SHORT launch sequence, independent of how fast this host enqueues: a link-bound stand-in launch
        (`sd_comm_sim_copy`, 16 MB at 4 GB/s = 4 ms) holds the stream while all n calls are enqueued behind it; stream events bracket the
        n calls.  (A decode at bs = 64 is ~21 us of device time for ~20 us of Python + ctypes + two launches: timed by the host clock it
        read 21.4-21.8 us on quiet boxes and 27.9 on a busy one.)"""
        from structuredetector_amd import _lib as L_
        if not _blk:
            _blk["src"] = torch.zeros(4 << 20, dtype=torch.float32, device=dev); _blk["dst"] = torch.empty_like(_blk["src"])
        for _ in range(warm):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        L_.check(L_.lib().sd_comm_sim_copy(_blk["src"].data_ptr(), _blk["dst"].data_ptr(), 16 << 20, 16 << 20, 8, 4.0, L_.stream()))
        e0.record()
        for _ in range(n):
            fn()
        e1.record(); e1.synchronize()
        return e0.elapsed_time(e1) * 1e-3 / n

    def guarded(ns, key, fn):
        """A side figure must never cost the headline line: errors are reported in place of the figure."""
        try:
            ns[key] = fn()
        except Exception as err:
            ns[key] = {"error": repr(err)[:300]}

    if rank == 0 and a.extras and world == 1:
        ns = {}

        def fig_fwd_fp32():
            # north_star target: ">= 60 % of the relevant roofline on the backbone forward at bs=64" (fp32 MFMA peak)
            with torch.no_grad():
                fwd = timed(lambda: net(images), 5)
            return {"batch": B, "ms": round(fwd * 1e3, 3), "tflops": round(B * FWD_GFLOP_PER_IMG / fwd / 1e3, 2),
                    "frac_of_fp32_mfma_peak": round(B * FWD_GFLOP_PER_IMG / fwd / 1e3 / PEAK_FP32_MFMA_TFLOPS, 4), "target_frac": 0.60}

        def fig_bs1():
            # BASELINE configs[1]: bs=1 inference = backbone forward + HIP decoder + host objects
            dec1 = Decoder(args)
            x1 = images[:1].contiguous()
            Mn = M + N
            with torch.no_grad():
                f1 = timed(lambda: net(x1), 20, warm=3)
                run1 = net.graphed(x1)
                run1.static_in.copy_(x1)
                g1 = timed(lambda: run1(run1.static_in), 20, warm=3)      # (the input already in the graph's static buffer: no copy launch)
                e2e = timed(lambda: dec1(net(x1)), 20, warm=3)

                def graph_e2e():
                    o = run1(x1)
                    return dec1({"anchor_hm": o[:, :M], "part_hm": o[:, M:Mn], "offsets": o[:, Mn:Mn + 2], "embeddings": o[:, Mn + 2:Mn + 4]})
                ge2e = timed(graph_e2e, 20, warm=3)
                del run1
                net.bf16_inference = True                      # the same image through the bf16 backbone (fp32 decode)
                net.invalidate_folded()
                try:
                    f1b = timed(lambda: net(x1), 20, warm=3)
                    e2eb = timed(lambda: dec1(net(x1)), 20, warm=3)
                    # the bf16 forward is ~0.42 ms of device time for 44 launches: slower host cores make the eager loop host-bound
                    # (~10 us of Python per launch); the replay is the device time
                    run1b = net.graphed(x1)
                    run1b.static_in.copy_(x1)
                    g1b = timed(lambda: run1b(run1b.static_in), 20, warm=3)
                    del run1b
                finally:
                    net.bf16_inference = False
                    net.invalidate_folded()
            return {"fwd_ms": round(f1 * 1e3, 3), "fwd_hipgraph_ms": round(g1 * 1e3, 3), "fwd_decode_objects_ms": round(e2e * 1e3, 3),
                    "hipgraph_fwd_decode_objects_ms": round(ge2e * 1e3, 3), "bf16_fwd_ms": round(f1b * 1e3, 3),
                    "bf16_fwd_hipgraph_ms": round(g1b * 1e3, 3), "bf16_fwd_decode_objects_ms": round(e2eb * 1e3, 3)}

        def measured_bf16_stream():
            # the box's own ceiling: a bare LDS-read + MFMA loop on random bf16 operands (csrc/sd_bench.hip), ~0.4 s of warm launches, the last ones timed
            if "tflops" not in bf16_stream:
                L_ = __import__("structuredetector_amd._lib", fromlist=["lib"])
                lib_ = L_.lib()
                gen_ = torch.Generator(device=dev).manual_seed(5)
                ops = (torch.rand(32 * 1024, device=dev, generator=gen_) * 2 - 1).to(torch.bfloat16)          # 64 KB of bf16 in [-1, 1)
                sink = torch.empty(256 * 512, dtype=torch.float32, device=dev)
                iters = 20000
                for _ in range(60):
                    L_.check(lib_.sd_mfma_bf16_stream(ops.data_ptr(), sink.data_ptr(), iters, L_.stream()))
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(5):
                    L_.check(lib_.sd_mfma_bf16_stream(ops.data_ptr(), sink.data_ptr(), iters, L_.stream()))
                e1.record(); e1.synchronize()
                bf16_stream["tflops"] = 5 * lib_.sd_mfma_bf16_stream_flops(iters) / (e0.elapsed_time(e1) * 1e-3) / 1e12
            return bf16_stream["tflops"]

        bf16_stream = {}

        def fig_fwd_bf16():
            # bf16 backbone (inference), same network object: bs=64 512x512, against the dense bf16 MFMA peak
            net.bf16_inference = True
            net.invalidate_folded()
            try:
                with torch.no_grad():
                    b16 = timed(lambda: net(images), 10, warm=3)
            finally:
                net.bf16_inference = False
                net.invalidate_folded()
            stream_tf = measured_bf16_stream()
            return {"batch": B, "ms": round(b16 * 1e3, 3), "tflops": round(B * FWD_GFLOP_PER_IMG / b16 / 1e3, 1),
                    "frac_of_bf16_mfma_peak": round(B * FWD_GFLOP_PER_IMG / b16 / 1e3 / PEAK_BF16_MFMA_TFLOPS, 4),
                    "measured_bf16_stream_tflops": round(stream_tf, 1),
                    "frac_of_measured_bf16_stream": round(B * FWD_GFLOP_PER_IMG / b16 / 1e3 / stream_tf, 4)}

        def fig_amp():
            # the same training step under `--amp` (trainer.py:115-121): bf16 activations / conv weights, fp32 accumulation + master weights
            step.amp = True
            try:
                t_amp = timed(lambda: run_step(0), 5, warm=2)
            finally:
                step.amp = False
            return {"batch": B, "ms_per_step": round(t_amp * 1e3, 3), "images_per_sec": round(B / t_amp, 1),
                    "speedup_vs_fp32_step": round((dt / a.steps) / t_amp, 2),
                    "note": "bf16 MFMA: stem conv and its weight gradient, forward, data-gradient, weight-gradient (transposed LDS reads), head forward; bf16 activations "
                            "everywhere between the stem conv and the head (stem tail and head backward included); fp32: all BatchNorm statistics / arithmetic, "
                            "accumulation, loss, Adam on fp32 master weights"}

        def fig_stress():
            # BASELINE configs[4]: 1024x1024, 8 labels / 8 parts, K=128, P=512, dense scenes (64-96 objects), bf16 backbone + fp32 decode
            Ms = Nn = 8; Ks, Ps, Bs, S = 128, 512, 16, 1024
            sargs = make_args(dev, Ms, Nn, Ks, Ps)
            sargs.use_amp = True
            snet = Network(sargs, pretrained=False).to(dev).eval()
            senc, sdec = Encode(sargs), Decoder(sargs)
            simg = torch.randn(Bs, 3, S, S, device=dev, generator=gen)
            stg = senc.render(senc.plan(S, S, *synthetic_batch(rng, Bs, S, S, Ms, Nn, 64, 96)), dev)
            shm = torch.cat([stg["anchor_hm"], stg["part_hm"]], 1).clamp(1e-4, 0.95)
            shead = torch.cat([torch.log(shm / (1 - shm)) + 0.05 * torch.randn(shm.shape, device=dev, generator=gen),
                               0.1 * torch.randn(Bs, 4, S // 4, S // 4, device=dev, generator=gen)], 1)
            souts = {"anchor_hm": shead[:, :Ms], "part_hm": shead[:, Ms:Ms + Nn], "offsets": shead[:, Ms + Nn:Ms + Nn + 2], "embeddings": shead[:, Ms + Nn + 2:]}
            with torch.no_grad():
                sf = timed(lambda: snet(simg), 5)
                sd_ = timed_device(lambda: sdec.decode_packed(souts, 0.5, 0.1, exact_topk=True), 10)
                sd_fast = timed_device(lambda: sdec.decode_packed(souts, 0.5, 0.1, exact_topk=False), 10)   # what Decoder.__call__ runs without metadata
                sboth = timed(lambda: (snet(simg), sdec.decode_packed(souts, 0.5, 0.1, exact_topk=True)), 5)
            return {"batch": Bs, "fwd_ms": round(sf * 1e3, 3), "fwd_tflops": round(Bs * STRESS_FWD_GFLOP_PER_IMG / sf / 1e3, 1),
                    "fwd_frac_of_bf16_mfma_peak": round(Bs * STRESS_FWD_GFLOP_PER_IMG / sf / 1e3 / PEAK_BF16_MFMA_TFLOPS, 4),
                    "fwd_frac_of_measured_bf16_stream": round(Bs * STRESS_FWD_GFLOP_PER_IMG / sf / 1e3 / measured_bf16_stream(), 4),
                    "decode_us_per_img": round(sd_ / Bs * 1e6, 2), "decode_GBps": round(Bs * STRESS_DECODE_BYTES_PER_IMG / sd_ / 1e9, 1),
                    "decode_frac_of_hbm_peak": round(Bs * STRESS_DECODE_BYTES_PER_IMG / sd_ / 1e9 / PEAK_HBM_GBPS, 4),
                    "decode_annotations_only_us_per_img": round(sd_fast / Bs * 1e6, 2),
                    "fwd_plus_decode_ms": round(sboth * 1e3, 3), "objects_per_img": "64-96", "K": Ks, "P": Ps}

        def fig_comm_sim():
            # What eight ranks will add to a step, sized on ONE GPU: a persistent copy launch of RCCL's shape (`--comm-sim-wgs` workgroups of 256
            # threads, 8 KB LDS, link-bound at `--comm-sim-gbps`) at the five bucket trigger points of the backward, on the exchange's side stream,
            # moving 2 x 7 / 8 x the bucket (`TrainStep.attach_sim`, csrc/sd_commsim.hip).  Same-process alternating A/B; `stretched` = the conv
            # kernels whose launches grew most per step (stream events around every conv launch, with and without the simulated exchange).
            def per_kernel(n_steps=2):
                ev = []
                net._engine.prof = ev
                for i in range(n_steps):
                    run_step(i)
                torch.cuda.synchronize()
                net._engine.prof = None
                agg = {}
                for kind, _fl, e0, e1, _ph in ev:
                    d = agg.setdefault(kind, [0, 0.0])
                    d[0] += 1; d[1] += e0.elapsed_time(e1)
                return {k: (v[0] / n_steps, v[1] / n_steps) for k, v in agg.items()}          # launches per step, ms per step

            out = {"ranks_modelled": 8, "workgroups": a.comm_sim_wgs, "bus_GBps_modelled": a.comm_sim_gbps,
                   "bytes_moved_per_step": int(2 * 7 / 8 * net.flat_grads.numel() * 4)}
            for label, amp in (("fp32", False), ("amp_bf16", True)):
                step.amp = amp
                try:
                    base, with_sim = [], []
                    for _ in range(3):
                        step.detach_sim()
                        base.append(timed(lambda: run_step(0), 4, warm=1))
                        step.attach_sim(ranks=8, workgroups=a.comm_sim_wgs, gbps=a.comm_sim_gbps)
                        with_sim.append(timed(lambda: run_step(0), 4, warm=1))
                    k_sim = per_kernel()
                    sim = step.sim
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    torch.cuda.synchronize(); e0.record(sim.side)
                    for name in step.STAGES:
                        sim.all_reduce(net.flat_grads, *step.ranges[name])
                    e1.record(sim.side); e1.synchronize()
                    isolated = e0.elapsed_time(e1)
                    step.detach_sim()
                    k_base = per_kernel()
                    grown = sorted(((k_sim[k][1] - k_base[k][1], k) for k in k_base if k in k_sim), reverse=True)[:3]
                    t0_, t1_ = min(base), min(with_sim)
                    out[label] = {"ms_per_step": round(t0_ * 1e3, 3), "ms_per_step_with_sim": round(t1_ * 1e3, 3),
                                  "exposed_ms": round((t1_ - t0_) * 1e3, 3), "exposed_pct": round(100 * (t1_ / t0_ - 1), 2),
                                  "sim_launches_isolated_ms": round(isolated, 3),
                                  "stretched": [{"kernel": k, "ms_per_step": round(k_base[k][1], 3), "ms_per_step_with_sim": round(k_sim[k][1], 3)} for _d, k in grown]}
                finally:
                    step.amp = False
                    step.detach_sim()
            return out

        net.eval()
        guarded(ns, "fwd_eval_fp32", fig_fwd_fp32)
        guarded(ns, "infer_bs1_fp32", fig_bs1)
        guarded(ns, "fwd_eval_bf16", fig_fwd_bf16)
        net.train()
        if not a.amp:
            guarded(ns, "train_step_amp_bf16", fig_amp)
            if a.comm_sim:
                guarded(ns, "comm_sim", fig_comm_sim)
        guarded(ns, "stress_1024_8x8_bf16", fig_stress)

        def fig_directory_feed():
            # `train --train_dir`: 512 PNG + JSON samples on disk -> decode threads -> pinned staging -> side-stream upload ->
            # GPU resize / ColorJitter / flips / Normalize / targets -> the same TrainStep; beside the synthetic-tensor step timed above
            import tempfile

            from tools.feed_bench import run as feed_run
            with tempfile.TemporaryDirectory(prefix="sd_feed_") as tmp:
                r = feed_run(n=512, batch=B, size=img, steps=16, amp=bool(a.amp), directory=tmp, breakdown=False, synthetic=False)
            r["synthetic_tensor_img_s"] = round(B * a.steps / dt, 1)
            r["directory_over_synthetic"] = round(r["directory_img_s"] / r["synthetic_tensor_img_s"], 3)
            return r
        if (B, img) == (64, 512):
            guarded(ns, "directory_feed", fig_directory_feed)
        extra["north_star"] = ns
    if rank == 0:
        try:
            dec = Decoder(args)
            tgt = enc.render_device(plans[0])
            hm = torch.cat([tgt["anchor_hm"], tgt["part_hm"]], 1).clamp(1e-4, 0.95)
            head = torch.cat([torch.log(hm / (1 - hm)) + 0.05 * torch.randn(hm.shape, device=dev, generator=gen),
                              0.1 * torch.randn(B, 4, img // 4, img // 4, device=dev, generator=gen)], 1)
            outs = {"anchor_hm": head[:, :M], "part_hm": head[:, M:M + N], "offsets": head[:, M + N:M + N + 2], "embeddings": head[:, M + N + 2:]}
            d_dev = timed_device(lambda: dec.decode_packed(outs, 0.5, 0.1, exact_topk=False), 20, warm=3)   # what Decoder.__call__ runs without metadata
            d_exact = timed_device(lambda: dec.decode_packed(outs, 0.5, 0.1, exact_topk=True), 20, warm=3)
            one = {k: v[:1] for k, v in outs.items()}
            d_one = timed(lambda: dec(one), 20, warm=3)                                             # launches + D2H + host assembly
            extra["decode"] = {"device_us_per_img_bs%d" % B: round(d_dev / B * 1e6, 3), "device_us_per_batch": round(d_dev * 1e6, 1),
                               "device_GBps_bs%d" % B: round(B * DECODE_BYTES_PER_IMG / d_dev / 1e9, 1),
                               "frac_of_hbm_peak": round(B * DECODE_BYTES_PER_IMG / d_dev / 1e9 / PEAK_HBM_GBPS, 4),
                               "exact_topk_us_per_img_bs%d" % B: round(d_exact / B * 1e6, 3),
                               "e2e_us_per_img_bs1": round(d_one * 1e6, 1), "bytes_per_img": DECODE_BYTES_PER_IMG,
                               # one-launch decodes that were redone because a selector block gave up its bounded wait (contention)
                               "selector_timeouts": int(dec.selector_timeouts)}
            if (B, img) == (64, 512):
                # the streaming rate at a batch that hides the fixed cost of the launches (bs = 512: 8 copies of the same head)
                big = {k: v.repeat(8, 1, 1, 1) for k, v in outs.items()}
                d_big = timed_device(lambda: dec.decode_packed(big, 0.5, 0.1, exact_topk=False), 20, warm=3)
                extra["decode"].update({"device_us_per_batch_bs512": round(d_big * 1e6, 1), "device_us_per_img_bs512": round(d_big / 512 * 1e6, 3),
                                        "device_GBps_bs512": round(512 * DECODE_BYTES_PER_IMG / d_big / 1e9, 1),
                                        "frac_of_hbm_peak_bs512": round(512 * DECODE_BYTES_PER_IMG / d_big / 1e9 / PEAK_HBM_GBPS, 4)})
                del big
            extra["decode_device_us_per_img_bs%d" % B] = extra["decode"]["device_us_per_img_bs%d" % B]     # (round-1 key names kept)
            extra["decode_device_GBps_bs%d" % B] = extra["decode"]["device_GBps_bs%d" % B]
            extra["decode_e2e_us_per_img_bs1"] = extra["decode"]["e2e_us_per_img_bs1"]
        except Exception as err:                      # never lose the headline line to a side figure
            extra["decode"] = {"error": repr(err)[:300]}

    if rank == 0:
        imgs_per_s = B * world * a.steps / dt
        line = {
            "metric": "images/sec (train fwd+bwd) and decode us/img at 512x512", "value": round(imgs_per_s, 2), "unit": "images/sec",
            "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": round(dt / a.steps * 1e3, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "bf16 (fp32 accumulate / master weights)" if a.amp else "f32",
            "data": "synthetic",
            "config": {"workload": f"configs[2]: train step bs={B}/GPU {img}x{img} fp32, 2 labels / 1 part, K=20 P=40, "
                                   "render targets + fwd + MSE/L1 loss + bwd + Adam; random-init ResNet-34+FPN",
                       "global_batch": B * world, "parallelism": f"dp{world}", "overlap_wgrad": bool(a.overlap_wgrad), "fuse_bn_bwd": bool(net._engine.fuse_bn_bwd), **({"zero_input": True} if a.zero_input else {}),
                       "exchange": "sd_allreduce (RCCL via C ABI)" if step.rccl is not None else (f"torch.distributed {dist.get_backend()}" if world > 1 else "none")},
            "train_tflops_per_gpu": round(B * TRAIN_GFLOP_PER_IMG * a.steps / dt / 1e3, 2),
            "train_frac_of_mfma_peak": round(B * TRAIN_GFLOP_PER_IMG * a.steps / dt / 1e3 / (PEAK_BF16_MFMA_TFLOPS if a.amp else PEAK_FP32_MFMA_TFLOPS), 4),
            "loss": [round(v, 6) for v in loss_host],
            "roofline": roofline,
            "rccl": rccl,
        }
        line.update(extra)
        if world == 1 and not a.no_cpu_baseline:
            try:
                line["cpu_baseline"] = cpu_baseline(M, N, K, P, img)
            except Exception as err:
                line["cpu_baseline"] = {"error": repr(err)[:300]}
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
