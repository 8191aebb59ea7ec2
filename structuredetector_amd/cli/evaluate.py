"""`evaluate` entry point (reference: src/sdnet/cli/evaluate.py:9-51): Network forward + Decoder over a validation
directory (or `--synthetic N` scenes), one image per decode call like the reference.  The reference's `Evaluator`
tables are outside the hot path; this prints a compact keypoint precision / recall / F1 from a greedy
nearest-ground-truth match within `--dist_threshold * min(width, height)`."""
import numpy as np
import torch

from ..data import CropDataset, Decoder, Encode
from ..model import Network
from ..utils import Arguments


def match(pred, truth, radius):
    """Greedy nearest match by descending score; pred: [(x, y, score)], truth: [(x, y)] -> true positives."""
    free = list(truth)
    tp = 0
    for (x, y, _s) in sorted(pred, key=lambda t: -t[2]):
        if not free:
            break
        d = [np.hypot(x - gx, y - gy) for gx, gy in free]
        j = int(np.argmin(d))
        if d[j] <= radius:
            tp += 1
            free.pop(j)
    return tp


def main(argv=None):
    args = Arguments().parse(argv)
    assert args.synthetic or args.valid_dir, "Path to a directory with validation samples must be specified."
    net = Network(args)
    if args.pretrained_model:
        net.load_state_dict(torch.load(args.pretrained_model, map_location="cpu"))
    net = net.eval().to(args.device)
    decoder = Decoder(args)
    radius = args.dist_threshold * min(args.width, args.height)
    stats = {"anchor": [0, 0, 0], "part": [0, 0, 0]}                     # tp, n_pred, n_truth

    def samples():
        if args.synthetic:
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
                yield torch.randn(3, args.height, args.width, device=args.device, generator=gen), ImageAnnotation(f"synthetic_{i}", objs)
        else:
            ds = CropDataset(args, args.valid_dir)
            for i in range(len(ds)):
                yield ds[i]

    for image, annotation in samples():
        with torch.no_grad():
            output = net(image[None].to(args.device))
        data = decoder(output, return_metadata=True)
        prediction = data["annotation"][0]
        stats["anchor"][0] += match([(o.x, o.y, o.anchor.score) for o in prediction.objects], [(o.x, o.y) for o in annotation.objects], radius)
        stats["anchor"][1] += len(prediction.objects); stats["anchor"][2] += len(annotation.objects)
        stats["part"][0] += match([(k.x, k.y, k.score) for k in data["raw_parts"][0]],
                                  [(p.x, p.y) for o in annotation.objects for p in o.parts], radius)
        stats["part"][1] += len(data["raw_parts"][0]); stats["part"][2] += annotation.nb_parts
    for name, (tp, npred, ntruth) in stats.items():
        prec, rec = tp / max(npred, 1), tp / max(ntruth, 1)
        print(f"{name:7s} precision {prec:.3f} recall {rec:.3f} f1 {2 * prec * rec / max(prec + rec, 1e-12):.3f} "
              f"(tp {tp}, predictions {npred}, ground truth {ntruth})")


if __name__ == "__main__":
    main()
