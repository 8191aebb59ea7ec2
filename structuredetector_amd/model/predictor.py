"""Batched inference over a dataset: the loop behind `evaluate`, the trainer's validation pass and `detect`.

The reference walks its validation set ONE image at a time (src/sdnet/cli/evaluate.py:34-45, model/trainer.py:137-168,
cli/detect.py:28-41): PIL decode + Resize + ToTensor + Normalize on the host, upload, forward, Decoder, Evaluator -- each step
waiting for the one before.  Images are independent units (decode / evaluate are replicas only), so here

  * only the image DECODE stays on the host (threads, `data/feeder.BatchFeeder`: pinned staging, side-stream upload);
  * Resize + Normalize run for the whole batch on the GPU (`sd_preprocess_images`: Pillow's fixed-point bilinear resampling and
    torchvision's to_tensor / Normalize arithmetic, bit-identical bytes -- tests/test_gpu_pipeline.py);
  * forward + decoder run at `--eval_batch` images per launch, and batch n+1 is QUEUED before batch n's packed result is read
    (`Decoder.submit` / `PendingDecode.result`), so the host's object assembly and metric code overlap the GPU;
  * results come back per image, in dataset order: `Evaluator.accumulate` sees the same sequence as the reference's walk.

`Predictor` mirrors src/sdnet/model/predictor.py:8-37 (one PIL image in, one ImageAnnotation out).
"""
from __future__ import annotations

import torch


def index_batches(n, batch):
    return [list(range(lo, min(lo + batch, n))) for lo in range(0, n, batch)]


def batched_outputs(net, decoder, dataset, args, batch=None, workers=None, with_raw_parts=True, keep_output=False, depth=2):
    """dataset: `CropDataset(args, dir, raw=True)` or `PredictionDataset(dir, args, raw=True)` -- items ((H, W, 3) uint8 tensor,
    ImageAnnotation in ORIGINAL pixels with img_size).  Yields, per image and in dataset order,
    (prediction ImageAnnotation in network-input pixels, ground-truth annotation in network-input pixels (resized + clipped like
    Resize + Encode do, transforms.py:47-60,154), raw_parts or None, this image's output dict of (1, C, h, w) views when keep_output else None)."""
    from ..data.augment import ValidationAugmentation
    from ..data.feeder import BatchFeeder, default_decode_workers
    batch = int(batch or getattr(args, "eval_batch", 16) or 16)
    workers = workers or getattr(args, "decode_workers", 0) or default_decode_workers()
    prepare = ValidationAugmentation(args)
    pending = None

    def finish(p):
        handle, anns, out = p
        preds, raws = handle.result()
        for i, (pred, ann) in enumerate(zip(preds, anns)):
            yield pred, ann, (raws[i] if raws is not None else None), (None if out is None else {k: v[i:i + 1] for k, v in out.items()})

    threads = torch.get_num_threads()
    torch.set_num_threads(1)                       # the loop's host work is tiny tensor ops; torch's pool would spin against the decode threads
    try:
        for group in BatchFeeder(dataset, index_batches(len(dataset), batch), args.device, workers=workers, depth=depth):
            with torch.no_grad():
                images, anns = prepare(group, group.annotations)
                out = net(images)
                if isinstance(out, torch.Tensor):  # Network(raw_output=True)
                    out = split_head(net, out)
                handle = decoder.submit(out, with_raw_parts=with_raw_parts)
            cur = (handle, anns, out if keep_output else None)
            if pending is not None:
                yield from finish(pending)
            pending = cur
        if pending is not None:
            yield from finish(pending)
    finally:
        torch.set_num_threads(threads)


def split_head(net, out):
    """The four channel-slice views `Network.forward` returns (network.py:77-84) of a raw head tensor."""
    M, nb = net.label_count, net.label_count + net.part_count
    return {"anchor_hm": out[:, :M], "part_hm": out[:, M:nb], "offsets": out[:, nb:nb + 2], "embeddings": out[:, nb + 2:nb + 4]}


class Predictor(torch.nn.Module):
    """src/sdnet/model/predictor.py:8-37: network (weights from `args.pretrained_model`) + Decoder behind one call;
    `forward(image)` takes a PIL image (any size) and returns its ImageAnnotation in network-input pixels."""

    def __init__(self, args):
        super().__init__()
        from ..data import Decoder
        from .network import Network
        self.args = args
        assert getattr(args, "pretrained_model", None), "No pretrained model specified. Use the option '--load_model <model_path>'."   # cli/evaluate.py:14-16
        self.model = Network(args, pretrained=False, init_weights=False)          # every tensor comes from the checkpoint (predictor.py:13-16)
        self.model.load_state_dict(torch.load(args.pretrained_model, map_location="cpu", weights_only=True), strict=True)
        self.model.eval().to(args.device)
        self.decoder = Decoder(args)

    def forward(self, image):
        import numpy as np

        from ..data.augment import preprocess_images
        arr = torch.from_numpy(np.asarray(image.convert("RGB"), np.uint8).copy())[None].to(self.args.device)
        with torch.no_grad():
            x = preprocess_images(arr, (self.args.width, self.args.height))
            return self.decoder(self.model(x))[0]
