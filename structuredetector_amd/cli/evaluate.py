"""`evaluate` entry point (reference: src/sdnet/cli/evaluate.py:9-51): Network forward + Decoder + Evaluator over a
validation directory (or `--synthetic N` seeded scenes); prints the reference's five metric tables and optionally writes the
keypoint CSV (`--save_csv_eval`).  The reference decodes one image per call (:34-45); here the directory is read by decode threads,
resized + normalised on the GPU and pushed through forward + decoder `--eval_batch` images at a time (model/predictor.py), while
`Evaluator.accumulate` still sees one image after the other in the reference's order -- counters and accuracy lists are identical."""
import torch

from ..data import CropDataset, Decoder
from ..model import Evaluator, Network
from ..utils import Arguments


def main(argv=None):
    args = Arguments().parse(argv)
    assert args.synthetic or args.valid_dir, "Path to a directory with validation samples must be specified."
    evaluator = Evaluator(args)
    decoder = Decoder(args)
    # evaluate.py:30-31 builds Network(args) (ImageNet trunk) and then overwrites every tensor from the checkpoint: the ImageNet file is
    # only looked up when there is no checkpoint to load
    net = Network(args, pretrained=not args.pretrained_model, init_weights=not args.pretrained_model)
    if args.pretrained_model:
        net.load_state_dict(torch.load(args.pretrained_model, map_location="cpu", weights_only=True))
    net = net.eval().to(args.device)
    if args.synthetic:
        from ..data.synthetic import synthetic_samples
        for image, annotation in synthetic_samples(args, args.synthetic):
            with torch.no_grad():
                output = net(image[None].to(args.device))
            data = decoder(output, return_metadata=True, metadata_fields=("annotation", "raw_parts"))
            evaluator.accumulate(data["annotation"][0], annotation, data["raw_parts"][0], True, True)
    else:
        from ..model.predictor import batched_outputs
        dataset = CropDataset(args, args.valid_dir, raw=True)               # decode only; Resize + Normalize run on the GPU per batch
        for prediction, annotation, raw_parts, _ in batched_outputs(net, decoder, dataset, args):
            # the annotation was resized to the network input and clipped (Resize + Encode's clip); the Evaluator maps both sides back to img_size
            evaluator.accumulate(prediction, annotation, raw_parts, True, True)
    evaluator.pretty_print()
    if args.csv_path is not None:
        evaluator.save_kps_csv(args.csv_path)
    return evaluator


if __name__ == "__main__":
    main()
