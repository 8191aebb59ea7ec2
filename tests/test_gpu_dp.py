"""Data-parallel TrainStep end to end on GPU tensors: 2 processes share the one GPU of the test box and exchange
gradients through gloo (RCCL cannot place two ranks on one device; the bucket / async / Adam-scaling logic is
backend-independent).  Checks: both ranks end with bit-identical weights, and the update equals Adam applied to the
MEAN of the two ranks' gradients computed independently."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _setup(rank_seed):
    from structuredetector_amd.data import Encode
    from structuredetector_amd.data.synthetic import synthetic_batch
    from structuredetector_amd.model import Network
    from tests.test_host_cpu import make_args
    dev = torch.device("cuda", 0)
    args = make_args(2, 1, 20, 40, device=dev, learning_rate=1e-3)
    torch.manual_seed(0)
    net = Network(args, pretrained=False).to(dev).train()
    enc = Encode(args)
    tgt = enc.render(enc.plan(128, 128, *synthetic_batch(np.random.default_rng(100 + rank_seed), 2, 128, 128, 2, 1)), dev)
    x = torch.randn(2, 3, 128, 128, device=dev, generator=torch.Generator(dev).manual_seed(200 + rank_seed))
    return args, net, x, tgt


def _worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from structuredetector_amd.model.trainer import TrainStep
        args, net, x, tgt = _setup(rank)
        step = TrainStep(net, args)
        assert step.world == 2
        step.sync_parameters()
        step(x, tgt)
        torch.cuda.synchronize()
        out[rank] = net.flat_params.cpu()
    finally:
        dist.destroy_process_group()


def test_two_rank_step_matches_mean_gradient_adam():
    from structuredetector_amd import _lib as L
    from structuredetector_amd.model.loss import loss_backward, loss_config, loss_forward
    world, port = 2, _free_port()
    out = mp.Manager().dict()
    mp.spawn(_worker, args=(world, port, out), nprocs=world, join=True)
    p0, p1 = out[0], out[1]
    assert torch.equal(p0, p1), "ranks diverged"
    # reference: gradients of each rank computed alone, averaged, one Adam step on rank 0's initial weights
    grads = []
    for r in range(world):
        args, net, x, tgt = _setup(r)
        head, tape = net.forward_train(x)
        cfg = loss_config(args, 2, 1, 20, 40)
        desc, keep, out8 = loss_forward(head, tgt, cfg)
        dhead = loss_backward(desc, out8, torch.ones((), device=head.device), tuple(head.shape))
        net.backward_from(tape, dhead)
        grads.append(net.flat_grads.clone())
        if r == 0:
            init = net.flat_params.clone()
    mean_g = ((grads[0] + grads[1]) * 0.5).contiguous()
    m = torch.zeros_like(init); v = torch.zeros_like(init)
    # grad_scale = 1/world is applied inside the kernel on the SUM, so feed the sum like TrainStep does
    gsum = (grads[0] + grads[1]).contiguous()
    L.check(L.lib().sd_adam_step(init.data_ptr(), gsum.data_ptr(), m.data_ptr(), v.data_ptr(), init.numel(), 1, 1e-3, 0.9, 0.999, 1e-8, 0.5, L.stream()))
    torch.cuda.synchronize()
    assert torch.equal(init.cpu(), p0), "DP step != Adam on the mean gradient"
    assert not torch.equal(mean_g, grads[0])


def _rccl_worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from structuredetector_amd.model.trainer import RcclExchange
        dev = torch.device("cuda", 0)
        ex = RcclExchange(dev)
        buf = torch.arange(1 << 20, dtype=torch.float32, device=dev) * 0.5
        ref = buf.clone()
        for lo, hi in ((0, 1000), (1000, 1 << 20)):
            ex.all_reduce(buf, lo, hi)
        ex.wait()
        torch.cuda.synchronize()
        out["equal"] = bool(torch.equal(buf, ref))
        ex.close()
    finally:
        dist.destroy_process_group()


def test_rccl_exchange_through_c_abi_single_rank():
    """The C-ABI RCCL wrapper on the one GPU of the test box: id bootstrap, communicator of one rank, bucketed in-place
    sums on the side stream, destroy.  (Two RCCL ranks need two devices: the driver's multi-GPU bench covers that.)
    Runs in a child process so that a communicator never lives in the pytest process."""
    out = mp.Manager().dict()
    mp.spawn(_rccl_worker, args=(1, _free_port(), out), nprocs=1, join=True)
    assert out["equal"] is True


@pytest.mark.parametrize("launcher", ["torchrun", "self"])
def test_bench_two_ranks_rehearsal_over_gloo(launcher):
    """bench.py's multi-rank path as the driver launches it (torch.distributed.run, one process per rank, barrier + max-over-ranks
    timing, rank 0 prints the one JSON line) with two ranks sharing the test GPU and gloo standing in for RCCL -- and the same run
    started as plain `python bench.py --gpus 2` (no launcher, WORLD_SIZE unset): bench.py then spawns its ranks itself before
    any GPU call and relays rank 0's line and the child's exit code."""
    import json
    import subprocess
    import sys
    from pathlib import Path
    root = Path(__file__).resolve().parent.parent
    env = dict(os.environ, SDNET_DIST_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT", "MASTER_ADDR"):
        env.pop(k, None)
    tail = [str(root / "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--batch", "4", "--size", "128"]
    if launcher == "torchrun":
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
               "--master-port", str(_free_port())] + tail
    else:
        cmd = [sys.executable] + tail
    out = subprocess.run(cmd, cwd=root, env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == 2 and rec["config"]["global_batch"] == 8 and rec["scaling"] == "weak" and rec["value"] > 0
    assert rec["config"]["exchange"] == "torch.distributed gloo" and "roofline" in rec
    # the line evidences its own exchange: every rank was in the sum, per-bucket durations, exposed communication time
    r = rec["rccl"]
    assert r["ranks"] == 2 and r["check"].startswith("ok: sum(rank+1) == 3") and r["backend"] == "gloo"
    assert set(r["buckets_isolated"]) == {"fpn_head", "down4", "down3", "down2", "down1_stem"}
    assert sum(b["floats"] for b in r["buckets_isolated"].values()) * 4 == r["allreduce_bytes_per_step"] > 87_000_000
    assert "exposed_comm_ms" in r and r["ms_per_step_without_exchange"] > 0


def test_verify_exchange_detects_a_missing_rank(monkeypatch):
    """TrainStep.verify_exchange on one rank reports 'no exchange'; with a world of 2 whose all-reduce does not happen (hooks
    stubbed out) the buffer holds rank+1 = 1 instead of sum(rank+1) = 3 and the check raises."""
    from structuredetector_amd import _lib as L
    from structuredetector_amd.model import trainer
    args, net, x, tgt = _setup(0)
    step = trainer.TrainStep(net, args)
    assert step.verify_exchange()["check"].startswith("single rank")
    step.world = 2
    monkeypatch.setattr(trainer.dist, "get_rank", lambda *a, **k: 0)
    monkeypatch.setattr(trainer.dist, "get_backend", lambda *a, **k: "stub")
    step._exchange_hooks = lambda: ((lambda name: None), (lambda: None))
    with pytest.raises(L.SdError, match="exchange check failed"):
        step.verify_exchange()


def _nccl_worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    torch.cuda.set_device(rank)
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", rank))
    try:
        from structuredetector_amd.data import Encode
        from structuredetector_amd.data.synthetic import synthetic_batch
        from structuredetector_amd.model import Network
        from structuredetector_amd.model.trainer import TrainStep
        from tests.test_host_cpu import make_args
        dev = torch.device("cuda", rank)
        args = make_args(2, 1, 20, 40, device=dev, learning_rate=1e-3)
        torch.manual_seed(0)
        net = Network(args, pretrained=False).to(dev).train()
        enc = Encode(args)
        tgt = enc.render(enc.plan(128, 128, *synthetic_batch(np.random.default_rng(100 + rank), 2, 128, 128, 2, 1)), dev)
        x = torch.randn(2, 3, 128, 128, device=dev, generator=torch.Generator(dev).manual_seed(200 + rank))
        res = {}
        for exchange in ("torch", "rccl"):
            step = TrainStep(net, args, exchange=exchange)
            step.sync_parameters()
            res[exchange + "_check"] = step.verify_exchange()["check"]
            step(x, tgt)
            torch.cuda.synchronize()
            res[exchange] = net.flat_params.cpu()
            if step.rccl is not None:
                step.rccl.close()
        out[rank] = res
    finally:
        dist.destroy_process_group()


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="RCCL needs one device per rank: runs on boxes with >= 2 GPUs")
def test_two_rank_step_over_rccl_when_two_gpus_are_present():
    """The real exchange (torch.distributed 'nccl' = RCCL, and the C-ABI sd_allreduce_* binding) with one rank per GPU: both
    self-checks pass and both ranks hold bit-identical weights after a step."""
    world, port = 2, _free_port()
    out = mp.Manager().dict()
    mp.spawn(_nccl_worker, args=(world, port, out), nprocs=world, join=True)
    for exchange in ("torch", "rccl"):
        assert out[0][exchange + "_check"].startswith("ok") and out[1][exchange + "_check"].startswith("ok")
        assert torch.equal(out[0][exchange], out[1][exchange]), f"ranks diverged with exchange={exchange}"


def test_simulated_exchange_keeps_the_step_and_takes_its_modelled_time():
    """`TrainStep(exchange="sim")` / `attach_sim` (what eight ranks add to a step, sized on one GPU): at the five bucket
    trigger points of the backward a persistent copy launch of RCCL's shape runs on the exchange's side stream (`sd_comm_sim_copy`:
    workgroups x 256 threads, link-bound at `gbps`).  Checked: (1) the step computes exactly what the single-rank step computes (parameters
    bit-identical after two steps: the copy goes to a scratch buffer, the mean over one rank is the gradient itself); (2) five launches
    move 2 x 7 / 8 x the flat gradient buffer; (3) a launch alone lasts move_bytes / gbps (the throttle, not the HBM, sets its duration) and
    copies what it should; (4) bad arguments are refused without a launch."""
    import ctypes as C
    from argparse import Namespace

    import numpy as np
    import torch

    from structuredetector_amd import _lib as L
    from structuredetector_amd.data import Encode
    from structuredetector_amd.data.synthetic import synthetic_batch
    from structuredetector_amd.model import Network
    from structuredetector_amd.model.trainer import TrainStep
    dev = torch.device("cuda")
    labels = {"a": 0, "b": 1}; parts = {"p": 0}
    args = Namespace(labels=labels, parts=parts, _r_labels={v: k for k, v in labels.items()}, _r_parts={v: k for k, v in parts.items()},
                     anchor_name="stem", down_ratio=4.0, max_objects=20, max_parts=40, conf_threshold=0.5, decoder_dist_thresh=0.1, sigma_gauss=0.1,
                     hm_loss_fn="mse", hm_weight=1.0, offset_weight=0.001, embedding_weight=0.001, fpn_depth=128, learning_rate=1e-3, device=dev)
    enc = Encode(args)
    x = torch.randn(2, 3, 128, 128, device=dev, generator=torch.Generator(device=dev).manual_seed(3))
    tgt = enc.render(enc.plan(128, 128, *synthetic_batch(np.random.default_rng(1), 2, 128, 128, 2, 1)), dev)
    finals = []
    for sim in (False, True):
        torch.manual_seed(11)
        net = Network(args, pretrained=False).to(dev).train()
        step = TrainStep(net, args)
        if sim:
            s = step.attach_sim(ranks=8, workgroups=16, gbps=400.0)
        for _ in range(2):
            step(x, tgt)
        torch.cuda.synchronize()
        finals.append(net.flat_params.clone())
        if sim:
            want = sum(int(2 * 7 / 8 * (hi - lo) * 4) // 16 * 16 for lo, hi in step.ranges.values())
            assert s.moved_bytes == 2 * want and want >= int(2 * 7 / 8 * net.flat_grads.numel() * 4) - 16 * 5
    assert torch.equal(finals[0], finals[1]), "the simulated exchange must not change what the step computes"
    lib = L.lib()
    n = 4 << 20                                                        # 16 MB at 100 GB/s: 168 us
    src = torch.arange(n, dtype=torch.float32, device=dev); dst = torch.zeros_like(src)
    for _ in range(2):
        L.check(lib.sd_comm_sim_copy(src.data_ptr(), dst.data_ptr(), n * 4, n * 4, 16, 100.0, L.stream()))
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    L.check(lib.sd_comm_sim_copy(src.data_ptr(), dst.data_ptr(), n * 4, n * 4, 16, 100.0, L.stream()))
    e1.record(); e1.synchronize()
    assert torch.equal(src, dst)
    us = e0.elapsed_time(e1) * 1e3
    # (lower bound: the throttle; upper bound generous -- the launch's own latency and a busy box add to it)
    assert 0.9 * 167.8 <= us <= 2.5 * 167.8, f"16 MB at a modelled 100 GB/s should take ~168 us, took {us:.1f}"
    assert lib.sd_comm_sim_copy(src.data_ptr(), dst.data_ptr(), n * 4, n * 4 + 8, 16, 100.0, L.stream()) == -1       # not a multiple of 16 bytes
    assert lib.sd_comm_sim_copy(src.data_ptr(), dst.data_ptr(), n * 4, n * 4, 0, 100.0, L.stream()) == -1           # no workgroups
    assert lib.sd_comm_sim_copy(src.data_ptr() + 4, dst.data_ptr(), n * 4 - 16, 16, 4, 100.0, L.stream()) == -3      # misaligned
