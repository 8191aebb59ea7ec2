"""Training step and loop for the SDNet hot path.

`TrainStep` is the inner step of the reference's `Trainer.train_epoch`
(src/sdnet/model/trainer.py:113-124: zero_grad, forward, loss, backward, Adam.step with lr 1e-3 and
default betas/eps, :53) as an explicit kernel schedule with no host synchronisation, extended with
what the reference lacks: pure data-parallel training, one process per GPU, gradients summed with
RCCL (`torch.distributed` backend "nccl") in five buckets that follow the backward pass
(FPN+head, down4, down3, down2, down1+stem), so the 87 MB exchange hides behind the remaining
backward kernels; the mean is applied inside the fused Adam launch (grad_scale = 1/world).
BatchNorm statistics stay rank-local (the reference has no SyncBN to match).
"""
from __future__ import annotations

import os

import torch
import torch.distributed as dist

from .. import _lib as L
from ..utils import trace as T
from .loss import LossStats, loss_backward, loss_config, loss_forward


BF16_TRAINING = True       # `--amp`: bf16 activations / conv weights, fp32 accumulation + master weights (autocast semantics of trainer.py:115-121)


class RcclExchange:
    """Gradient sum straight through the C ABI (`sd_allreduce_*`, RCCL): what a host without torch.distributed would
    call.  The 128-byte RCCL id travels over the already initialised process group's store (any host channel works);
    each bucket's all-reduce runs on a side stream that first waits for the compute stream (so it is ordered after the
    weight-gradient kernels issued so far) and `wait()` joins it back before the Adam launch.  Opt-in
    (`TrainStep(..., exchange="rccl")` or SDNET_EXCHANGE=rccl); the default exchange is torch.distributed's RCCL binding."""

    def __init__(self, device, process_group=None):
        import ctypes as C
        rank, world = dist.get_rank(process_group), dist.get_world_size(process_group)
        ident = C.create_string_buffer(128)
        if rank == 0:
            L.check(L.lib().sd_allreduce_unique_id(ident), "sd_allreduce_unique_id")
        box = [ident.raw if rank == 0 else None]
        dist.broadcast_object_list(box, src=0, group=process_group)
        ident = C.create_string_buffer(box[0], 128)
        self._comm = C.c_void_p()
        with torch.cuda.device(device):
            L.check(L.lib().sd_allreduce_init(ident, rank, world, C.byref(self._comm)), "sd_allreduce_init")
        self.world = world
        self.side = torch.cuda.Stream(device)

    def all_reduce(self, flat, lo, hi):
        self.side.wait_stream(torch.cuda.current_stream())
        L.check(L.lib().sd_allreduce_run(self._comm, flat.data_ptr() + 4 * lo, hi - lo, self.side.cuda_stream), "sd_allreduce_run")

    def wait(self):
        torch.cuda.current_stream().wait_stream(self.side)

    def close(self):
        if self._comm:
            L.check(L.lib().sd_allreduce_destroy(self._comm), "sd_allreduce_destroy")
            self._comm = None


class SimExchange:
    """Stand-in for the gradient all-reduce on ONE GPU (`TrainStep(..., exchange="sim")`): at every bucket trigger point of the backward a
    persistent copy launch of RCCL's shape (`sd_comm_sim_copy`: `workgroups` blocks of 256 threads, 8 KB of LDS, link-bound at `gbps`) moves
    2 (N - 1) / N x the bucket (N = `ranks`) on the exchange's side stream -- same ordering as `RcclExchange` (waits for the compute stream,
    joined before Adam).  The gradients are NOT changed (the copy goes to a scratch buffer): the step computes what a single rank computes;
    only its timing carries the collective's footprint.  bench.py: `north_star.comm_sim`."""

    def __init__(self, device, largest_bucket_floats, ranks=8, workgroups=32, gbps=200.0):
        self.ranks, self.workgroups, self.gbps = int(ranks), int(workgroups), float(gbps)
        self.side = torch.cuda.Stream(device)
        self.scratch = torch.empty(largest_bucket_floats, dtype=torch.float32, device=device)
        self.moved_bytes = 0

    def all_reduce(self, flat, lo, hi):
        n = hi - lo
        if n > self.scratch.numel():
            raise L.SdError(f"SimExchange: bucket of {n} floats exceeds the scratch buffer ({self.scratch.numel()})")
        move = int(2 * (self.ranks - 1) / self.ranks * n * 4) // 16 * 16
        self.side.wait_stream(torch.cuda.current_stream())
        L.check(L.lib().sd_comm_sim_copy(flat.data_ptr() + 4 * lo, self.scratch.data_ptr(), n * 4, move, self.workgroups, self.gbps, self.side.cuda_stream),
                "sd_comm_sim_copy")
        self.moved_bytes += move

    def wait(self):
        torch.cuda.current_stream().wait_stream(self.side)


class TrainStep:
    def __init__(self, net, args, lr=None, betas=(0.9, 0.999), eps=1e-8, process_group=None, exchange=None):
        if net.flat_params is None:
            raise L.SdError("move the Network to the GPU before building TrainStep")
        self.net, self.args = net, args
        self.lr = float(lr if lr is not None else getattr(args, "learning_rate", 1e-3))
        self.betas, self.eps = betas, eps
        self.exp_avg = torch.zeros_like(net.flat_params)
        self.exp_avg_sq = torch.zeros_like(net.flat_params)
        self.step_count = 0
        self.pg = process_group
        self.world = dist.get_world_size(process_group) if (dist.is_available() and dist.is_initialized()) else 1
        exchange = exchange or os.environ.get("SDNET_EXCHANGE", "torch")
        want_sim = exchange == "sim"
        if exchange not in ("torch", "rccl", "sim"):
            raise ValueError(f"exchange must be 'torch', 'rccl' or 'sim', got {exchange!r}")
        self.rccl = RcclExchange(net.flat_params.device, process_group) if (exchange == "rccl" and self.world > 1) else None
        self.ranges = net.stage_ranges()
        self.sim = None                  # SimExchange: the collective's footprint on one GPU (attach_sim)
        if want_sim:
            self.attach_sim()
        self.one = torch.ones((), dtype=torch.float32, device=net.flat_params.device)
        self.stats = LossStats()
        self.amp = bool(getattr(args, "use_amp", False))      # mixed-precision step (trainer.py:115-121)
        self.exchange_enabled = True     # False: skip the all-reduce (bench.py measures the exposed communication time with it)

    # ---- true resume: the reference saves weights only (trainer.py:226-237), so a run cannot continue --------
    def state_dict(self):
        """Optimizer state of the flat-buffer Adam: both moment buffers, the step count (bias correction) and the learning rate.
        Tensors are copies on the host; `Network.state_dict()` carries the weights and BatchNorm buffers."""
        return {"exp_avg": self.exp_avg.detach().cpu().clone(), "exp_avg_sq": self.exp_avg_sq.detach().cpu().clone(),
                "step_count": int(self.step_count), "lr": float(self.lr), "betas": tuple(self.betas), "eps": float(self.eps),
                "flat_numel": int(self.net.flat_params.numel())}

    def load_state_dict(self, state):
        if int(state["flat_numel"]) != self.net.flat_params.numel():
            raise L.SdError(f"optimizer state is for a flat parameter buffer of {state['flat_numel']} floats, this network has "
                            f"{self.net.flat_params.numel()} (different labels / parts / fpn_depth?)")
        self.exp_avg.copy_(state["exp_avg"])
        self.exp_avg_sq.copy_(state["exp_avg_sq"])
        self.step_count, self.lr = int(state["step_count"]), float(state["lr"])
        self.betas, self.eps = tuple(state["betas"]), float(state["eps"])

    def sync_parameters(self):
        """Identical initial weights on every rank (rank 0's)."""
        if self.world > 1:
            dist.broadcast(self.net.flat_params, 0, group=self.pg)
            for b in self.net.buffers():
                dist.broadcast(b, 0, group=self.pg)

    # Bucket plans: which stages travel together.  A group is launched when its LAST stage completes; the stages of a group are adjacent in
    # the flat buffer (parameters are laid out in registration order: stem, down1 .. down4, FPN, head -- the backward completes them from
    # the end), so a group is ONE contiguous all-reduce.  Five launches hide best under a 73 ms fp32 step; a 15 ms mixed-precision step pays
    # ~0.08 ms of fixed cost per launch (stream hand-offs on both sides + the launch itself: `profiles/r05_comm_sim_sweep.txt`).
    BUCKET_PLANS = {5: (("fpn_head",), ("down4",), ("down3",), ("down2",), ("down1_stem",)),
                    3: (("fpn_head", "down4"), ("down3",), ("down2", "down1_stem")),
                    2: (("fpn_head", "down4", "down3"), ("down2", "down1_stem")),
                    1: (("fpn_head", "down4", "down3", "down2", "down1_stem"),)}

    def set_bucket_plan(self, launches):
        """Number of all-reduce launches per step: 5 (default: one per parameter group), 3, 2 or 1."""
        if launches not in self.BUCKET_PLANS:
            raise ValueError(f"bucket plan must be one of {sorted(self.BUCKET_PLANS)}, got {launches!r}")
        self.bucket_plan = launches

    def _groups(self):
        """{trigger stage: (lo, hi)} of the current plan: the flat range a group covers, keyed by the stage that completes it."""
        out = {}
        for group in self.BUCKET_PLANS[getattr(self, "bucket_plan", 5)]:
            lo = min(self.ranges[n][0] for n in group); hi = max(self.ranges[n][1] for n in group)
            assert hi - lo == sum(self.ranges[n][1] - self.ranges[n][0] for n in group), "the stages of a bucket group must be adjacent in the flat buffer"
            out[group[-1]] = (lo, hi)
        return out

    def _exchange_hooks(self):
        """(on_stage, finish): on_stage(name) starts the all-reduce of the gradient bucket group that stage `name` completes (called by the
        backward schedule as soon as that parameter group's gradients are complete), finish() joins every launch before the Adam launch."""
        net = self.net
        groups = self._groups()
        sim = getattr(self, "sim", None)
        if sim is not None and self.exchange_enabled:
            return (lambda name: sim.all_reduce(net.flat_grads, *groups[name]) if name in groups else None), sim.wait
        if self.world == 1 or not self.exchange_enabled:
            return None, (lambda: None)
        if self.rccl is not None:
            return (lambda name: self.rccl.all_reduce(net.flat_grads, *groups[name]) if name in groups else None), self.rccl.wait
        works = []

        def on_stage(name):
            if name in groups:
                lo, hi = groups[name]
                works.append(dist.all_reduce(net.flat_grads[lo:hi], group=self.pg, async_op=True))

        def finish():
            for w in works:
                w.wait()
        return on_stage, finish

    STAGES = ("fpn_head", "down4", "down3", "down2", "down1_stem")       # order in which the backward completes the buckets

    def attach_sim(self, ranks=8, workgroups=32, gbps=200.0):
        """Run every following step with the simulated exchange (`SimExchange`) beside its backward; `detach_sim()` ends it."""
        if self.world > 1:
            raise L.SdError("attach_sim replaces the gradient exchange by a stand-in that moves no gradients: single-rank sizing runs only "
                            f"(this step has {self.world} ranks)")
        # (scratch as large as the whole flat buffer: a bucket plan may send several parameter groups -- up to all of them -- in one launch)
        self.sim = SimExchange(self.net.flat_params.device, int(self.net.flat_grads.numel()), ranks, workgroups, gbps)
        return self.sim

    def detach_sim(self):
        self.sim = None

    def verify_exchange(self):
        """Self-check of the data-parallel exchange THROUGH THE PATH THE STEP USES (same buckets, same binding, same streams):
        every rank fills its flat gradient buffer with rank+1, the five buckets are summed, and every element must read
        world*(world+1)/2 on every rank.  Returns the report bench.py prints; raises if a rank is missing from the sum."""
        net, n = self.net, self.world
        report = {"ranks": n, "exchange": self.exchange_name(), "buckets": {k: self.ranges[k][1] - self.ranges[k][0] for k in self.STAGES}}
        if n == 1:
            report["check"] = "single rank: no exchange"
            return report
        rank = dist.get_rank(self.pg)
        net.flat_grads.fill_(float(rank + 1))
        on_stage, finish = self._exchange_hooks()
        for name in self.STAGES:
            on_stage(name)
        finish()
        want = n * (n + 1) / 2
        lo, hi = float(net.flat_grads.min()), float(net.flat_grads.max())
        net.flat_grads.zero_()
        if lo != want or hi != want:
            raise L.SdError(f"gradient exchange check failed on rank {rank}: sum of (rank+1) over {n} ranks should be {want}, got [{lo}, {hi}]")
        report["check"] = f"ok: sum(rank+1) == {want:g} in all {len(self.STAGES)} buckets on every rank"
        return report

    def exchange_name(self):
        if self.world == 1:
            return "none"
        if self.rccl is not None:
            return "sd_allreduce (RCCL via C ABI)"
        return f"torch.distributed {dist.get_backend(self.pg)}"

    def time_buckets(self, iters=5):
        """Isolated duration of each bucket's all-reduce in microseconds (stream events around a blocking all-reduce of the
        bucket, nothing else running; median of `iters`): what the overlap with the backward kernels has to hide."""
        if self.world == 1:
            return {}
        net, out = self.net, {}
        for name in self.STAGES:
            lo, hi = self.ranges[name]
            ts = []
            for _ in range(iters + 1):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                if self.rccl is not None:
                    self.rccl.all_reduce(net.flat_grads, lo, hi); self.rccl.wait()
                else:
                    dist.all_reduce(net.flat_grads[lo:hi], group=self.pg)
                e1.record(); e1.synchronize()
                ts.append(e0.elapsed_time(e1) * 1e3)
            ts = sorted(ts[1:])
            out[name] = {"floats": hi - lo, "us": round(ts[len(ts) // 2], 1),
                         "algbw_GBps": round((hi - lo) * 4 / (ts[len(ts) // 2] * 1e-6) / 1e9, 1)}
        net.flat_grads.zero_()
        return out

    def __call__(self, images, targets):
        """One optimizer step; returns the loss vector [total, hm, offset, embedding] as a device tensor."""
        net = self.net
        T.push("step")
        T.push("forward")
        head, tape = net.forward_train(images, amp=self.amp)
        T.pop()
        T.push("loss")
        M, N = net.label_count, net.part_count
        cfg = loss_config(self.args, M, N, targets["anchor_inds"].shape[1], targets["part_inds"].shape[1])
        desc, keep, out8 = loss_forward(head, targets, cfg)
        dhead = loss_backward(desc, out8, self.one, tuple(head.shape))
        T.pop()
        on_stage, finish = self._exchange_hooks()
        if on_stage is not None and T.enabled():
            launch = on_stage

            def on_stage(name):                                    # every gradient bucket as its own range
                T.push("bucket:" + name)
                launch(name)
                T.pop()
        T.push("backward")
        net.backward_from(tape, dhead, on_stage)
        T.pop()
        T.push("exchange:join")
        finish()
        T.pop()
        self.step_count += 1
        T.push("adam")
        L.check(L.lib().sd_adam_step(net.flat_params.data_ptr(), net.flat_grads.data_ptr(), self.exp_avg.data_ptr(),
                                     self.exp_avg_sq.data_ptr(), net.flat_params.numel(), self.step_count, self.lr, self.betas[0],
                                     self.betas[1], self.eps, 1.0 / self.world, L.stream()), "sd_adam_step")
        T.pop()
        T.pop()
        self.stats.update(out8[1], out8[2], out8[3])
        return out8[:4]


class StepLR:
    """torch.optim.lr_scheduler.StepLR(step_size, gamma=0.1) over TrainStep.lr (trainer.py:54-56)."""

    def __init__(self, step: TrainStep, step_size: int, gamma: float = 0.1):
        self.step_obj, self.step_size, self.gamma, self.epoch, self.base = step, max(int(step_size), 1), gamma, 0, step.lr

    def step(self):
        self.epoch += 1
        self.step_obj.lr = self.base * self.gamma ** (self.epoch // self.step_size)

    def state_dict(self):
        return {"epoch": self.epoch, "base": self.base, "step_size": self.step_size, "gamma": self.gamma}

    def load_state_dict(self, state):
        self.epoch, self.base, self.step_size, self.gamma = int(state["epoch"]), float(state["base"]), int(state["step_size"]), float(state["gamma"])
        self.step_obj.lr = self.base * self.gamma ** (self.epoch // self.step_size)


def shard_indices(n, batch, rank, world, seed):
    """Per-rank batches of sample indices for one epoch of data-parallel training over a real dataset.
    Every rank draws the SAME permutation (shared seed = base seed + epoch), truncates it to whole global batches
    (drop_last=True, trainer.py:69, applied to the global batch `batch * world`) and takes the strided shard
    `[rank::world]`: shards are disjoint, cover the truncated permutation and have the same number of steps on every
    rank, so the bucketed all-reduces of all ranks pair up step by step."""
    import numpy as np
    order = np.random.default_rng(seed).permutation(n)
    usable = (n // (batch * world)) * batch * world
    mine = order[:usable][rank::world]
    return [mine[i:i + batch] for i in range(0, len(mine), batch)]


class Trainer:
    """Epoch loop around TrainStep with the reference's knobs (src/sdnet/model/trainer.py:23-237): Adam(lr),
    StepLR(step_size=args.lr_step), validation every second epoch with the Decoder + Evaluator + Loss and the four
    `model_best_{loss,csi,classif,kp_reg}.pth` checkpoints in trainings/<timestamp>/.  Data: a directory of JSON+image
    samples (no augmentation) or `--synthetic N` seeded scenes rendered on the GPU.  TensorBoard logging and the PIL
    augmentations are outside the hot path."""

    def __init__(self, args):
        from datetime import datetime
        from pathlib import Path

        import numpy as np

        from ..data import CropDataset, Encode
        from .network import Network
        self.args = args
        if getattr(args, "use_amp", False) and not BF16_TRAINING:
            # the reference autocasts the training forward under --amp (trainer.py:115-121); silently training in fp32 under the
            # same flag would be a different experiment with the same name
            raise NotImplementedError("--amp: the bf16 training step is not built; train without --amp (fp32), or use "
                                      "--bf16_inference to run only the validation / inference forward on the bf16 backbone")
        self.rank = dist.get_rank() if dist.is_initialized() else 0
        self.epoch = 0
        self.net = Network(args, pretrained=True)
        if args.pretrained_model:
            self.net.load_state_dict(torch.load(args.pretrained_model, map_location="cpu", weights_only=True))
        self.net.to(args.device).train()
        self.step = TrainStep(self.net, args, lr=args.learning_rate)
        self.step.sync_parameters()
        self.scheduler = StepLR(self.step, args.lr_step)
        self.encode = Encode(args)
        self.rng = np.random.default_rng(926354916 + self.rank)
        # directory data: decode on the host (PIL), resize + flips + normalisation for the whole batch on the GPU
        from ..data.augment import TrainAugmentation
        self.dataset = None if args.synthetic else CropDataset(args, args.train_dir, raw=True)
        self.augment = TrainAugmentation(args)
        self.save_dir = Path("trainings") / f"{datetime.now():%Y-%m-%d_%H-%M-%S}"
        self.best_loss = float("inf")
        self.best_csi = self.best_classif = self.best_kp_reg = 0.0
        from ..data import Decoder
        from .evaluator import Evaluator
        from .loss import Loss
        self.decoder, self.evaluator, self.loss = Decoder(args), Evaluator(args), Loss(args)
        self.valid_set = None if args.synthetic or not args.valid_dir else CropDataset(args, args.valid_dir, raw=True)
        self.start_epoch = 0
        self.global_step = 0                      # trainer.py:35,129: images seen, the x axis of every scalar
        self._writer = None
        if getattr(args, "resume", None):
            self.load_resume(args.resume)

    def batches(self):
        a, B = self.args, self.args.batch_size
        world = self.step.world
        if a.synthetic:
            from ..data.synthetic import synthetic_batch
            gen = torch.Generator(device=a.device).manual_seed(926354916 + self.rank)
            for _ in range(max(a.synthetic // (B * world), 1)):
                flat = synthetic_batch(self.rng, B, a.width, a.height, len(a.labels), len(a.parts))
                images = torch.randn(B, 3, a.height, a.width, device=a.device, generator=gen)
                yield images, self.encode.render(self.encode.plan(a.width, a.height, *flat), a.device)
        else:
            # decode threads + pinned staging + side-stream upload, `prefetch` batches ahead of the step (data/feeder.py); resize, jitter,
            # flips, normalisation and the targets for the whole batch on the GPU (the reference: trainer.py:62-72, dataset.py:41-49)
            from ..data.feeder import BatchFeeder, default_decode_workers
            shards = shard_indices(len(self.dataset), B, self.rank, world, 926354916 + self.epoch)
            workers = getattr(a, "decode_workers", 0) or default_decode_workers(world)
            # While the feed runs, torch's intra-op pool is single-threaded: the loop's host work is thousands of tiny tensor ops and kernel
            # launches, the pool gives them nothing and its workers spin against the decode threads (measured: 64 x 7 scalar draws 450 ms
            # next to 16 decoders, 7 ms alone).  The reference's DataLoader workers run single-threaded for the same reason.
            threads = torch.get_num_threads()
            torch.set_num_threads(1)
            try:
                for batch in BatchFeeder(self.dataset, shards, a.device, workers=workers, depth=getattr(a, "prefetch", 3)):
                    images, anns = self.augment(batch, batch.annotations)
                    yield images, self.encode.batch(self.augment.size, anns, a.device)
            finally:
                torch.set_num_threads(threads)

    def valid_samples(self):
        """Per image, in order: (prediction, annotation in network-input pixels, raw_parts, this image's head views)."""
        a = self.args
        if self.valid_set is not None:
            # decode threads -> GPU Resize + Normalize -> forward + decoder at --eval_batch images per launch (model/predictor.py)
            from .predictor import batched_outputs
            yield from batched_outputs(self.net, self.decoder, self.valid_set, a, keep_output=True)
            return
        from ..data.synthetic import synthetic_samples
        for image, annotation in synthetic_samples(a, min(max(a.synthetic, 1), 16), seed=20261003):
            with torch.no_grad():
                output = self.net(image[None].to(a.device))
                data = self.decoder(output, return_metadata=True, metadata_fields=("annotation", "raw_parts"))
            yield data["annotation"][0], annotation, data["raw_parts"][0], output

    def valid(self):
        """Validation pass of the reference (src/sdnet/model/trainer.py:137-237): eval-mode forward, Decoder + Evaluator + Loss PER IMAGE
        (the loss statistics are means over images of per-image losses, :160-168), then the four `model_best_*.pth` checkpoints
        (rank 0 only).  Forward and decoder run batched; the per-image loss is taken on that image's slice of the batch output."""
        a = self.args
        self.net.eval()
        self.evaluator.reset()
        stats, n, per_image = LossStats(), 0, []
        for prediction, annotation, raw_parts, output in self.valid_samples():
            with torch.no_grad():
                self.evaluator.accumulate(prediction, annotation, raw_parts, eval_csi=True, eval_classif=True)
                # the annotation is in network-input pixels (resized + clipped / the synthetic generator): encode it as the target
                target = self.encode.batch((a.width, a.height), [annotation], a.device)
                self.loss(output, target)
            per_image.append(torch.stack([self.loss.stats.hm_loss, self.loss.stats.offset_loss, self.loss.stats.embedding_loss]))
            n += 1
        if n:                                                      # one host sync for the whole pass (it was three `.item()` per image)
            stats = LossStats(*(torch.stack(per_image).double().sum(0).tolist()))
        self.net.train()
        if n:
            stats /= n
        f1_csi = self.evaluator.csi_eval.reduce().f1_score
        f1_classif = self.evaluator.classification_eval.reduce().f1_score
        f1_kp = self.evaluator.kps_eval.reduce().f1_score
        if self.rank == 0:
            self.save_dir.mkdir(parents=True, exist_ok=True)
            from ..utils.scalars import metric_dicts
            writer = self.scalar_writer()
            writer.add_scalars("Loss/Validation", dict(hm_loss=stats.hm_loss, offset_loss=stats.offset_loss,
                                                        embedding_loss=stats.embedding_loss), self.global_step)      # trainer.py:240-242
            for tag, values in metric_dicts(self.evaluator).items():                                                 # trainer.py:243-256
                writer.add_scalars(tag, values, self.global_step)
            writer.flush()
            print(f"validation ({n} images): loss {stats.total_loss:.5f} | kp F1 {f1_kp:.2%} | CSI F1 {f1_csi:.2%} | classification F1 {f1_classif:.2%}", flush=True)
            for value, attr, name, better in ((stats.total_loss, "best_loss", "loss", lambda v, b: v < b), (f1_csi, "best_csi", "csi", lambda v, b: v > b),
                                              (f1_classif, "best_classif", "classif", lambda v, b: v > b), (f1_kp, "best_kp_reg", "kp_reg", lambda v, b: v > b)):
                if better(value, getattr(self, attr)):
                    setattr(self, attr, value)
                    self.net.save(self.save_dir / f"model_best_{name}.pth")
        return stats

    # ---- true resume: weights + BatchNorm buffers + Adam moments / step + StepLR epoch + best-so-far metrics + epoch counter ----
    def save_resume(self, path):
        """Everything the next epoch depends on: weights + BatchNorm buffers, Adam moments / step / lr, StepLR epoch, best-so-far
        metrics, the multi-scale input size drawn for the next epoch (trainer.py:135), the random streams the augmentation and the
        synthetic scenes draw from (torch's global generator, this rank's numpy generator) and the run's save directory -- tensors
        and plain Python values only, so that the file loads with `weights_only=True`."""
        torch.save({"model": {k: v.detach().cpu().clone() for k, v in self.net.state_dict().items()},
                    "optimizer": self.step.state_dict(), "scheduler": self.scheduler.state_dict(), "epoch": self.epoch,
                    "best": {k: getattr(self, k) for k in ("best_loss", "best_csi", "best_classif", "best_kp_reg")},
                    "augment_size": tuple(int(v) for v in self.augment.size), "torch_rng": torch.get_rng_state(),
                    "numpy_rng": self.rng.bit_generator.state, "save_dir": str(self.save_dir), "global_step": int(self.global_step)}, path)

    def load_resume(self, path):
        from pathlib import Path
        state = torch.load(path, map_location="cpu", weights_only=True)
        self.net.load_state_dict(state["model"])
        self.step.load_state_dict(state["optimizer"])
        self.scheduler.load_state_dict(state["scheduler"])
        self.start_epoch = int(state["epoch"]) + 1
        for k, v in state["best"].items():
            setattr(self, k, v)
        if "augment_size" in state:                     # (files written before round 3 carry the weights / optimizer part only)
            self.augment.size = tuple(int(v) for v in state["augment_size"])
            torch.set_rng_state(state["torch_rng"])
            self.rng.bit_generator.state = state["numpy_rng"]
            self.save_dir = Path(state["save_dir"])     # model_best_* and resume.pth of one run stay in one directory
        self.global_step = int(state.get("global_step", 0))

    def scalar_writer(self):
        """The run's scalar log (rank 0), created on first use: TensorBoard when importable, always `<dir>/scalars.jsonl` (utils/scalars.py)."""
        if self._writer is None:
            from ..utils.scalars import ScalarWriter
            self._writer = ScalarWriter(getattr(self.args, "log_dir", None) or self.save_dir) if self.rank == 0 else ScalarWriter(None)
        return self._writer

    def train(self):
        try:
            self._train()
        finally:                                                       # the TensorBoard event file and scalars.jsonl are closed with the run
            if self._writer is not None:
                self._writer.close()
                self._writer = None

    def _train(self):
        steps = 0
        for epoch in range(self.start_epoch, self.args.epochs):
            self.epoch = epoch
            per_step = []
            for images, targets in self.batches():
                per_step.append(self.step(images, targets).clone())    # (total, hm, offset, embedding) on the device: no host sync in the loop
                steps += 1
                if self.args.steps and steps >= self.args.steps:
                    break
            n = len(per_step)
            rows = torch.stack(per_step).tolist() if n else []         # one host sync per epoch
            mean = [sum(r[k] for r in rows) / max(n, 1) for k in range(4)]
            if self.rank == 0:
                # trainer.py:126-133: "Loss/Train" per optimizer step at global_step (+= batch_size per step), "Learning rate" per epoch;
                # the reference's writer forces a device sync per step -- the same scalars are written here from the epoch's one copy
                writer = self.scalar_writer()
                for r in rows:
                    writer.add_scalars("Loss/Train", dict(hm_loss=r[1], offset_loss=r[2], embedding_loss=r[3]), self.global_step)
                    self.global_step += self.args.batch_size
                writer.add_scalar("Learning rate", self.step.lr, self.global_step)
                writer.flush()
                print(f"epoch {epoch}: total {mean[0]:.5f} hm {mean[1]:.5f} offset {mean[2]:.5f} embedding {mean[3]:.5f} "
                      f"lr {self.step.lr:g} ({n} steps)", flush=True)
            if epoch % 2 == 0:                                         # trainer.py:98-99
                self.valid()
            self.scheduler.step()
            self.augment.trigger_random_resize()                       # trainer.py:135: a new input size (multiple of 32) per epoch
            if self.rank == 0:                                         # state at the END of the epoch: --resume continues with epoch + 1
                self.save_dir.mkdir(parents=True, exist_ok=True)
                self.save_resume(self.save_dir / "resume.pth")
            if self.args.steps and steps >= self.args.steps:
                break
