"""`evaluate` entry point (reference: src/sdnet/cli/evaluate.py:9-51): Network forward + Decoder + Evaluator over a
validation directory (or `--synthetic N` seeded scenes), one image per decode call like the reference; prints the
reference's five metric tables and optionally writes the keypoint CSV (`--save_csv_eval`)."""
import numpy as np
import torch

from ..data import CropDataset, Decoder
from ..model import Evaluator, Network
from ..utils import Arguments


def _synthetic_samples(args):
    from ..data.synthetic import synthetic_batch
    from ..utils import ImageAnnotation, Keypoint, Object
    rng = np.random.default_rng(926354916)
    gen = torch.Generator(device=args.device).manual_seed(926354916)
    for i in range(args.synthetic):
        n_obj, o_lab, o_xy, o_np, p_kind, p_xy = synthetic_batch(rng, 1, args.width, args.height, len(args.labels), len(args.parts))
        objs, j = [], 0
        for k in range(int(n_obj[0])):
            parts = [Keypoint(args._r_parts[int(p_kind[j + q])], *p_xy[j + q]) for q in range(int(o_np[k]))]
            j += int(o_np[k])
            objs.append(Object(args._r_labels[int(o_lab[k])], Keypoint(args.anchor_name, *o_xy[k]), parts))
        ann = ImageAnnotation(f"synthetic_{i}", objs, img_size=(args.width, args.height))
        yield torch.randn(3, args.height, args.width, device=args.device, generator=gen), ann


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
        samples = _synthetic_samples(args)
    else:
        ds = CropDataset(args, args.valid_dir)
        samples = (ds[i] for i in range(len(ds)))
    for image, annotation in samples:
        with torch.no_grad():
            output = net(image[None].to(args.device))
        data = decoder(output, return_metadata=True)
        # CropDataset resized the annotation to the network input; the Evaluator maps both sides back to img_size
        evaluator.accumulate(data["annotation"][0], annotation, data["raw_parts"][0], True, True)
    evaluator.pretty_print()
    if args.csv_path is not None:
        evaluator.save_kps_csv(args.csv_path)
    return evaluator


if __name__ == "__main__":
    main()
