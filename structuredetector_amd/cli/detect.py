"""`detect` entry point (reference: src/sdnet/cli/detect.py:13-53): run the network + decoder over every `.jpg` of
`--valid_dir`, write `predictions/<name>.json` (annotation in original image pixels) and the image with the objects drawn.
Images are decoded by threads and go through Resize + Normalize, forward and decoder in batches of `--eval_batch`
(model/predictor.py); outputs are written per image in the directory's order, like the reference's walk."""
from pathlib import Path

import torch
from PIL import Image

from ..data import Decoder
from ..data.dataset import PredictionDataset
from ..model import Network
from ..utils import Arguments, draw


def main(argv=None):
    args = Arguments().parse(argv)
    assert args.valid_dir, "Path to a directory with the images to process must be specified (--valid_dir)."
    dataset = PredictionDataset(args.valid_dir, args, raw=True)
    decoder = Decoder(args)
    net = Network(args, pretrained=not args.pretrained_model, init_weights=not args.pretrained_model)             # detect.py:24-25: every tensor comes from the checkpoint
    if args.pretrained_model:
        net.load_state_dict(torch.load(args.pretrained_model, map_location="cpu", weights_only=True))
    net = net.eval().to(args.device)
    out_dir = Path("predictions")
    out_dir.mkdir(exist_ok=True)
    written = []
    from ..model.predictor import batched_outputs
    for annotation, source, _, _ in batched_outputs(net, decoder, dataset, args, with_raw_parts=False):
        img_size, image_path = source.img_size, source.image_path
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
