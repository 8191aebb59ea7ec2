"""`detect` entry point (reference: src/sdnet/cli/detect.py:13-53): run the network + decoder over every `.jpg` of
`--valid_dir`, write `predictions/<name>.json` (annotation in original image pixels) and the image with the objects drawn."""
from pathlib import Path

import numpy as np
import torch
from PIL import Image

from ..data import Decoder
from ..data.dataset import PredictionDataset
from ..model import Network
from ..utils import Arguments, draw


def main(argv=None):
    args = Arguments().parse(argv)
    assert args.valid_dir, "Path to a directory with the images to process must be specified (--valid_dir)."
    dataset = PredictionDataset(args.valid_dir, args)
    decoder = Decoder(args)
    net = Network(args)
    if args.pretrained_model:
        net.load_state_dict(torch.load(args.pretrained_model, map_location="cpu"))
    net = net.eval().to(args.device)
    out_dir = Path("predictions")
    out_dir.mkdir(exist_ok=True)
    written = []
    from ..data.feeder import prefetch_items
    for item, image_path in zip(prefetch_items(dataset, getattr(args, "decode_workers", 0) or None), dataset.images):
        with torch.no_grad():
            output = net(item["img"][None].to(args.device))
        img_size = item["img_size"]
        annotation = decoder(output)[0]
        annotation.resize((args.width, args.height), img_size)          # back to the pixels of the original image
        annotation.img_size = img_size
        annotation.image_path = image_path
        image = draw(Image.open(image_path).convert("RGB"), annotation, args)
        annotation.save_json(out_dir)
        image.save(out_dir / image_path.name)
        written.append(out_dir / image_path.with_suffix(".json").name)
    return written


detect = main

if __name__ == "__main__":
    main()
