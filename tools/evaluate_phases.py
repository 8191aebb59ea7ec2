#!/usr/bin/env python3
"""Where `evaluate --valid_dir` spends its wall time (second pass, warm): argument parsing, Network construction, upload, the batched
loop (feeder wait / preprocess + forward + decoder submit / result + accumulate per batch), report.
usage: evaluate_phases.py [--n 256] [--eval_batch 16] [--workers 0]"""
import argparse
import contextlib
import io
import sys
import time
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from tools.feed_bench import write_samples  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=256); ap.add_argument("--eval_batch", type=int, default=16); ap.add_argument("--workers", type=int, default=0)
    ap.add_argument("--dir", default="/tmp/sd_eval")
    a = ap.parse_args()
    from structuredetector_amd.data import CropDataset, Decoder
    from structuredetector_amd.data.augment import ValidationAugmentation
    from structuredetector_amd.data.feeder import BatchFeeder, default_decode_workers
    from structuredetector_amd.model import Evaluator, Network
    from structuredetector_amd.model.predictor import index_batches
    from structuredetector_amd.utils import Arguments
    root = Path(a.dir)
    labels = write_samples(root / "valid", a.n, 512)
    argv = ["--valid_dir", str(root / "valid"), "--labels", str(labels), "-s", "stem", "--eval_batch", str(a.eval_batch)]
    for rep in range(2):
        T = {}
        t0 = time.perf_counter(); args = Arguments().parse(argv); T["parse"] = time.perf_counter() - t0
        t0 = time.perf_counter(); net = Network(args, pretrained=False); T["Network()"] = time.perf_counter() - t0
        t0 = time.perf_counter(); net = net.eval().to(args.device); torch.cuda.synchronize(); T["to(device)"] = time.perf_counter() - t0
        ev, dec, ds = Evaluator(args), Decoder(args), CropDataset(args, args.valid_dir, raw=True)
        prep = ValidationAugmentation(args)
        workers = a.workers or default_decode_workers()
        wait = launch = finish = acc = 0.0
        t_loop = time.perf_counter()
        pending = None
        torch.set_num_threads(1)
        it = iter(BatchFeeder(ds, index_batches(len(ds), a.eval_batch), args.device, workers=workers, depth=2))
        while True:
            t0 = time.perf_counter()
            group = next(it, None)
            t1 = time.perf_counter(); wait += t1 - t0
            if group is not None:
                with torch.no_grad():
                    images, anns = prep(group, group.annotations)
                    out = net(images)
                    handle = dec.submit(out, with_raw_parts=True)
                cur = (handle, anns)
            t2 = time.perf_counter(); launch += t2 - t1
            if pending is not None:
                preds, raws = pending[0].result()
                t3 = time.perf_counter(); finish += t3 - t2
                for p, g, r in zip(preds, pending[1], raws):
                    ev.accumulate(p, g, r, True, True)
                acc += time.perf_counter() - t3
            if group is None:
                break
            pending = cur
        T["loop"] = time.perf_counter() - t_loop
        t0 = time.perf_counter()
        with contextlib.redirect_stdout(io.StringIO()):
            ev.pretty_print()
        T["report"] = time.perf_counter() - t0
    tot = sum(T.values())
    print(f"n={a.n} eval_batch={a.eval_batch} decode workers={workers}: total {tot * 1e3:.0f} ms = {tot / a.n * 1e3:.2f} ms per image")
    print("  " + ", ".join(f"{k} {v * 1e3:.0f} ms" for k, v in T.items()))
    print(f"  loop per image: feeder wait {wait / a.n * 1e3:.2f} ms, preprocess + forward + decoder launches {launch / a.n * 1e3:.2f} ms, "
          f"result wait + assembly {finish / a.n * 1e3:.2f} ms, Evaluator.accumulate {acc / a.n * 1e3:.2f} ms")


if __name__ == "__main__":
    main()
