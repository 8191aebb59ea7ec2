"""`evaluate` entry point (reference: src/sdnet/cli/evaluate.py:9-51): Network forward + Decoder + Evaluator over a
validation directory (or `--synthetic N` seeded scenes), one image per decode call like the reference; prints the
reference's five metric tables and optionally writes the keypoint CSV (`--save_csv_eval`)."""
import torch

from ..data import CropDataset, Decoder
from ..model import Evaluator, Network
from ..utils import Arguments


def main(argv=None):
    args = Arguments().parse(argv)
    assert args.synthetic or args.valid_dir, "Path to a directory with validation samples must be specified."
    evaluator = Evaluator(args)
    decoder = Decoder(args)
    net = Network(args)
    if args.pretrained_model:
        net.load_state_dict(torch.load(args.pretrained_model, map_location="cpu"))
    net = net.eval().to(args.device)
    if args.synthetic:
        from ..data.synthetic import synthetic_samples
        samples = synthetic_samples(args, args.synthetic)
    else:
        from ..data.feeder import prefetch_items
        ds = CropDataset(args, args.valid_dir)
        samples = prefetch_items(ds, getattr(args, "decode_workers", 0) or None)      # same items, same order; decode runs ahead on threads
    for image, annotation in samples:
        with torch.no_grad():
            output = net(image[None].to(args.device))
        data = decoder(output, return_metadata=True, metadata_fields=("annotation", "raw_parts"))
        # CropDataset resized the annotation to the network input; the Evaluator maps both sides back to img_size
        evaluator.accumulate(data["annotation"][0], annotation, data["raw_parts"][0], True, True)
    evaluator.pretty_print()
    if args.csv_path is not None:
        evaluator.save_kps_csv(args.csv_path)
    return evaluator


if __name__ == "__main__":
    main()
