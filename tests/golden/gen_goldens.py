"""Generate golden vectors by running the REAL reference on seeded inputs.

Run in the build container only (needs /root/reference):

    PYTHONPATH=/root/reference/src PYTHONDONTWRITEBYTECODE=1 python tests/golden/gen_goldens.py

The reference is imported from where it lies (nothing is copied); torchvision and
tensorboard are absent from the container, so empty in-memory module stubs are
registered for them before the import -- none of the functions
exercised here touches those modules.  Outputs: tests/golden/*.npz (inputs +
expected outputs, data only).  Version skew: reference pins torch 2.5.1, this
container runs the torch recorded in each file's ``meta``.
"""
import json
import re
import sys
import types
from argparse import Namespace
from pathlib import Path

import numpy as np
import torch

HERE = Path(__file__).resolve().parent
ROOT = HERE.parent.parent
sys.path.insert(0, str(ROOT))

for name in ["torchvision", "torchvision.transforms", "torchvision.transforms.functional", "torchvision.models"]:
    sys.modules[name] = types.ModuleType(name)
sys.modules["torchvision"].transforms = sys.modules["torchvision.transforms"]
sys.modules["torchvision.transforms"].functional = sys.modules["torchvision.transforms.functional"]
sys.modules["torchvision"].models = sys.modules["torchvision.models"]
sys.modules["torchvision.models"].resnet34 = None
sys.modules["torchvision.models"].ResNet34_Weights = None
_tb = types.ModuleType("torch.utils.tensorboard")
_tb.SummaryWriter = object
sys.modules["torch.utils.tensorboard"] = _tb

import sdnet.utils as RU  # noqa: E402  (the reference)
from sdnet.data.dataset import CropDataset  # noqa: E402
from sdnet.data.decoders import Decoder  # noqa: E402
from sdnet.data.transforms import Encode  # noqa: E402
from sdnet.model.evaluator import Evaluator  # noqa: E402
from sdnet.model.loss import Loss  # noqa: E402
from sdnet.model.network import Fpn, Head  # noqa: E402

from oracle import sdnet_oracle as O  # noqa: E402  (only for the synthetic INPUT generator)

META = json.dumps({"torch": torch.__version__, "numpy": np.__version__, "reference_pins": "torch 2.5.1"})


def make_args(M, N, K, P, hm_loss_fn="mse", anchor_name="stem"):
    labels = {f"label{i}": i for i in range(M)}
    parts = {f"part{i}": i for i in range(N)}
    return Namespace(labels=labels, parts=parts, _r_labels={v: k for k, v in labels.items()},
                     _r_parts={v: k for k, v in parts.items()}, anchor_name=anchor_name, down_ratio=4.0,
                     max_objects=K, max_parts=P, conf_threshold=0.5, decoder_dist_thresh=0.1, sigma_gauss=0.1,
                     hm_loss_fn=hm_loss_fn, hm_weight=1.0, offset_weight=0.001, embedding_weight=0.001)


def to_annotation(args, objs, name="img.png"):
    out = []
    for (label, x, y, parts) in objs:
        kps = [RU.Keypoint(args._r_parts[k], px, py) for (k, px, py) in parts]
        out.append(RU.Object(args._r_labels[label], RU.Keypoint(args.anchor_name, x, y), kps))
    return RU.ImageAnnotation(name, out)


def flat_scene(objs):
    """(label,x,y,parts) list -> flat float64 arrays (data only)."""
    o = np.array([[l, x, y, len(p)] for (l, x, y, p) in objs], np.float64).reshape(-1, 4)
    p = np.array([[k, px, py] for (_, _, _, ps) in objs for (k, px, py) in ps], np.float64).reshape(-1, 3)
    return o, p


def annotation_to_arrays(args, ann):
    """ImageAnnotation -> objs (n,4) [label_idx,x,y,score], parts (m,5) [obj,kind_idx,x,y,score] float64."""
    objs, parts = [], []
    for oi, obj in enumerate(ann.objects):
        objs.append([args.labels[obj.name], obj.anchor.x, obj.anchor.y, obj.anchor.score])
        for kp in obj.parts:
            parts.append([oi, args.parts[kp.kind], kp.x, kp.y, kp.score])
    return np.array(objs, np.float64).reshape(-1, 4), np.array(parts, np.float64).reshape(-1, 5)


def check_margins(scores_sorted, what, rel_gap=4e-6):
    """tie-free guard: consecutive positive scores must differ by >= ~32 fp32 ulps (relative),
    so that 1-2 ulp differences between sigmoid implementations cannot reorder them."""
    s = np.asarray(scores_sorted, np.float64)
    pos = s[s > 0]
    gaps = -np.diff(pos) / pos[1:]
    assert (gaps > rel_gap).all(), f"{what}: near-tie in golden input (min relative gap {gaps.min()})"


# --------------------------------------------------------------------------
def gen_prims(rng):
    out = {"meta": META}
    x = (3.0 * rng.standard_normal((2, 3, 40, 56))).astype(np.float32)
    x[0, 0, 5, 5] = 30.0; x[0, 0, 5, 7] = 31.0   # saturated -> clamp plateau (both survive: equal after clamp)
    x[1, 2, 0, 0] = 9.0; x[1, 2, 39, 55] = 8.5    # borders / corners
    t = torch.from_numpy(x)
    sig = RU.clamped_sigmoid(t)
    out["logits"] = x
    out["sig"] = sig.numpy()
    out["nms"] = RU.nms(sig).numpy()
    # topk on a tie-free dense map (no NMS): random distinct values
    d = rng.permutation(2 * 3 * 40 * 56).astype(np.float32).reshape(2, 3, 40, 56) / 16384.0
    out["dense"] = d
    for k in (2, 7, 40):  # k=1 trips the squeeze(-1) bug at utils.py:343
        s, i, c, y, xx = RU.topk(torch.from_numpy(d), k=k)
        out[f"topk{k}_score"] = s.numpy(); out[f"topk{k}_ind"] = i.numpy(); out[f"topk{k}_cls"] = c.numpy()
        out[f"topk{k}_y"] = y.numpy(); out[f"topk{k}_x"] = xx.numpy()
    feat = rng.standard_normal((2, 2, 40, 56)).astype(np.float32)
    ind = rng.integers(0, 40 * 56, (2, 9))
    out["feat"] = feat; out["gind"] = ind
    out["gathered"] = RU.transpose_and_gather(torch.from_numpy(feat), torch.from_numpy(ind)).numpy()
    v = (100 * rng.standard_normal((4, 5, 6, 2))).astype(np.float32)
    out["hyp_in"] = v; out["hyp_out"] = RU.hypot(torch.from_numpy(v)).numpy()
    Y, X = torch.meshgrid(torch.arange(32), torch.arange(48), indexing="ij")
    out["gauss"] = RU.gaussian_2d(X, Y, 17, 5, 0.1 * 32 / 3).numpy()
    np.savez_compressed(HERE / "prims.npz", **out)


def gen_encode_decode(rng, tag, img, M, N, K, P, n_img, n_min, n_max, noise):
    args = make_args(M, N, K, P)
    enc_ref = Encode(args); dec_ref = Decoder(args)
    out = {"meta": META, "cfg": np.array([img, img, M, N, K, P], np.int64), "noise": np.float64(noise)}
    samples, heads = [], []
    for n in range(n_img):
        objs = O.synthetic_scene(rng, img, img, M, N, n_min, n_max)
        if n == 1:
            objs = objs[:1]                      # near-empty scene
        if n == 2:
            objs = []                            # empty scene
        so, sp = flat_scene(objs)
        out[f"scene{n}_objs"] = so; out[f"scene{n}_parts"] = sp
        e = enc_ref(torch.zeros(3, img, img), to_annotation(args, objs))
        for k in ["anchor_hm", "part_hm", "anchor_inds", "part_inds", "anchor_offsets", "part_offsets",
                  "embeddings", "anchor_mask", "part_mask"]:
            out[f"enc{n}_{k}"] = e[k].numpy()
        samples.append(e)
        enp = {k: (v.numpy() if isinstance(v, torch.Tensor) else v) for k, v in e.items()}
        heads.append(O.head_from_targets(rng, enp, M, N, noise=noise))
    head = np.stack(heads, 0)
    out["head"] = head
    th = torch.from_numpy(head)
    outputs = {"anchor_hm": th[:, :M], "part_hm": th[:, M:M + N], "offsets": th[:, M + N:M + N + 2],
               "embeddings": th[:, M + N + 2:]}
    md = dec_ref(outputs, return_metadata=True)
    names = ["score", "ind", "cls", "y", "x"]
    for nm, v in zip(names, md["topk_anchor"]):
        out[f"dec_anchor_{nm}"] = v.numpy()
    for nm, v in zip(names, md["topk_kp"]):
        out[f"dec_part_{nm}"] = v.numpy()
    out["dec_embeddings"] = md["embeddings"].numpy()
    for b in range(n_img):
        check_margins(np.sort(RU.nms(md["anchor_hm_sig"])[b].numpy().ravel())[::-1][:K + 1], f"{tag} anchors img{b}")
        check_margins(np.sort(RU.nms(md["part_hm_sig"])[b].numpy().ravel())[::-1][:P + 1], f"{tag} parts img{b}")
        o, p = annotation_to_arrays(args, md["annotation"][b])
        out[f"ann{b}_objs"] = o; out[f"ann{b}_parts"] = p
        out[f"raw{b}"] = np.array([[args.parts[k.kind], k.x, k.y, k.score] for k in md["raw_parts"][b]],
                                  np.float64).reshape(-1, 4)
    # loss goldens on the same (head, collated targets), mse and focal, value + autograd grad
    batch = CropDataset.collate_fn([dict(s, image=torch.zeros(1), annotation=None) for s in samples])
    for fn in ("mse", "focal"):
        largs = make_args(M, N, K, P, hm_loss_fn=fn)
        lref = Loss(largs)
        x = th.clone().requires_grad_(True)
        od = {"anchor_hm": x[:, :M], "part_hm": x[:, M:M + N], "offsets": x[:, M + N:M + N + 2], "embeddings": x[:, M + N + 2:]}
        val = lref(od, batch)
        val.backward()
        out[f"loss_{fn}"] = np.array([float(val), float(lref.stats.hm_loss), float(lref.stats.offset_loss),
                                      float(lref.stats.embedding_loss)], np.float64)
        out[f"lossgrad_{fn}"] = x.grad.numpy()
    np.savez_compressed(HERE / f"{tag}.npz", **out)
    print(tag, "objects per image:", [len(md["annotation"][b].objects) for b in range(n_img)])


def gen_truncation(rng):
    """K / P truncation quirk of Encode (transforms.py:157,186-191)."""
    M, N, K, P, img = 2, 2, 4, 6, 128
    args = make_args(M, N, K, P)
    out = {"meta": META, "cfg": np.array([img, img, M, N, K, P], np.int64)}
    cases = {
        "many_objs": O.synthetic_scene(rng, img, img, M, N, 7, 7, 1, 1),       # > K objects
        "many_parts": O.synthetic_scene(rng, img, img, M, N, 3, 3, 3, 3),      # parts hit P before objects run out
        "exact_parts": O.synthetic_scene(rng, img, img, M, N, 4, 4, 2, 2)[:3] + O.synthetic_scene(rng, img, img, M, N, 1, 1, 1, 1),
        "out_of_bounds": [(0, -5.0, 140.0, [(1, 300.0, -2.0)]), (1, 127.0, 127.0, [(0, 126.99, 0.0)])],
    }
    out["cases"] = np.array(list(cases))
    for name, objs in cases.items():
        so, sp = flat_scene(objs)
        out[f"{name}_objs"] = so; out[f"{name}_parts"] = sp
        e = Encode(args)(torch.zeros(3, img, img), to_annotation(args, objs))
        for k in ["anchor_hm", "part_hm", "anchor_inds", "part_inds", "anchor_offsets", "part_offsets",
                  "embeddings", "anchor_mask", "part_mask"]:
            out[f"{name}_{k}"] = e[k].numpy()
    np.savez_compressed(HERE / "encode_trunc.npz", **out)


def gen_evaluator(rng):
    """Reference Evaluator on decoded scenes (labels bean / maize / leaf so that the hard-coded classification labels apply)."""
    M, N, K, P, img = 2, 1, 20, 40, 512
    args = make_args(M, N, K, P)
    args.labels = {"bean": 0, "maize": 1}; args.parts = {"leaf": 0}
    args._r_labels = {0: "bean", 1: "maize"}; args._r_parts = {0: "leaf"}
    args.width = args.height = img; args.dist_threshold = 0.05; args.csi_threshold = 0.75
    ev = Evaluator(args)
    dec_ref, enc_ref = Decoder(args), Encode(args)
    out = {"meta": META, "cfg": np.array([img, img, M, N, K, P], np.int64)}
    n_img = 6
    for n in range(n_img):
        objs = O.synthetic_scene(rng, img, img, M, N, 4, 10, 0, 3)
        ann = to_annotation(args, objs); ann.img_size = (img + 64 * (n % 2), img)     # also exercises the resize to the image size
        e = enc_ref(torch.zeros(3, img, img), to_annotation(args, objs))
        enp = {k: (v.numpy() if isinstance(v, torch.Tensor) else v) for k, v in e.items()}
        head = torch.from_numpy(O.head_from_targets(rng, enp, M, N, noise=0.6, reg_noise=0.5))[None]   # noisy: FPs, FNs, bad links
        md = dec_ref({"anchor_hm": head[:, :M], "part_hm": head[:, M:M + N], "offsets": head[:, M + N:M + N + 2],
                      "embeddings": head[:, M + N + 2:]}, return_metadata=True)
        pred, raw = md["annotation"][0], md["raw_parts"][0]
        so, sp = flat_scene(objs)
        out[f"gt{n}_objs"] = so; out[f"gt{n}_parts"] = sp; out[f"gt{n}_size"] = np.array(ann.img_size, np.int64)
        po, pp = annotation_to_arrays(args, pred)
        out[f"pred{n}_objs"] = po; out[f"pred{n}_parts"] = pp
        out[f"raw{n}"] = np.array([[args.parts[k.kind], k.x, k.y, k.score] for k in raw], np.float64).reshape(-1, 4)
        ev.accumulate(pred, ann, raw, True, True)
    for sec, evals in (("anchor", ev.anchor_eval), ("part", ev.part_eval), ("csi", ev.csi_eval), ("classif", ev.classification_eval)):
        out[f"{sec}_labels"] = np.array(list(evals.labels))
        out[f"{sec}_counts"] = np.array([[e.tp, e.npos, e.ndet] for _, e in evals.items()], np.int64)
        for label, e in evals.items():
            out[f"{sec}_acc_{label}"] = np.array(e.acc, np.float64)
    out["csv"] = np.array(ev._csv_kps_str())
    np.savez_compressed(HERE / "evaluator.npz", **out)
    print("evaluator:", ev.anchor_eval.reduce(), "|", ev.csi_eval.reduce())


def _store_decode(out, md, args, n_img):
    names = ["score", "ind", "cls", "y", "x"]
    for nm, v in zip(names, md["topk_anchor"]):
        out[f"dec_anchor_{nm}"] = v.numpy()
    for nm, v in zip(names, md["topk_kp"]):
        out[f"dec_part_{nm}"] = v.numpy()
    out["dec_embeddings"] = md["embeddings"].numpy()
    for b in range(n_img):
        o, p = annotation_to_arrays(args, md["annotation"][b])
        out[f"ann{b}_objs"] = o; out[f"ann{b}_parts"] = p
        out[f"raw{b}"] = np.array([[args.parts[k.kind], k.x, k.y, k.score] for k in md["raw_parts"][b]],
                                  np.float64).reshape(-1, 4)


def gen_thresholds(rng):
    """Threshold edge cases of Decoder.__call__ (decoders.py:78,83,100,115-117,153) with thresholds that
    fp32 cannot represent: conf 0.4 (fp32 0.4000000060) and dist 0.1 * 64 = 6.4 (fp32 6.4000000954).
      A1 / P2: score == fp32(0.4) exactly -> masked on the device (fp32 `>`), yet the anchor is still emitted as a
               part-less object (double(score) > 0.4) and the part stays in raw_parts (double(score) < 0.4 is False);
      P0: origin exactly fp32(6.4) from anchor A0 -> NOT attached (fp32 `<`); P1: one ulp closer -> attached;
      P3: sits on the masked anchor A1 -> unattached (A1 is at the +1e6 sentinel); P4: ordinary attachment to A2."""
    M, N, K, P, hw = 2, 1, 6, 8, 64
    args = make_args(M, N, K, P)
    conf, dist = 0.4, 0.1
    f32 = np.float32
    c32 = f32(conf)
    x0 = f32(np.log(0.4 / 0.6))
    cands = [x0]
    for _ in range(40):
        cands.append(np.nextafter(cands[-1], f32(1)))
    lo = x0
    for _ in range(40):
        lo = np.nextafter(lo, f32(-1)); cands.append(lo)
    cands = np.array(sorted(cands), f32)
    sig = RU.clamped_sigmoid(torch.from_numpy(cands)).numpy()
    hits = cands[sig == c32]
    assert len(hits), "no fp32 logit whose reference sigmoid is exactly fp32(0.4)"
    l04 = hits[len(hits) // 2]
    head = np.zeros((1, M + N + 4, hw, hw), f32)
    head[0, :M + N] = -8.0 + 0.3 * rng.standard_normal((M + N, hw, hw)).astype(f32)
    d32 = f32(dist * hw)
    below = np.nextafter(d32, f32(0))
    anchors = [(0, 0, 10, 3.0), (1, 30, 30, l04), (0, 50, 50, 2.0), (1, 20, 50, -1.0)]          # (class, x, y, logit)
    parts = [(8, 10, 2.5, d32 - f32(8), f32(0)), (8, 16, 2.4, below - f32(8), f32(-6)), (40, 30, l04, f32(0), f32(0)),
             (31, 36, 2.2, f32(-1), f32(-6)), (52, 44, 2.1, f32(-2), f32(6))]                  # (x, y, logit, ex, ey)
    for (c, x, y, lg) in anchors:
        head[0, c, max(y - 3, 0):y + 4, max(x - 3, 0):x + 4] = -12.0
        head[0, c, y, x] = lg
    for (x, y, lg, ex, ey) in parts:
        head[0, M, max(y - 3, 0):y + 4, max(x - 3, 0):x + 4] = -12.0
        head[0, M, y, x] = lg
        head[0, M + N + 2, y, x] = ex; head[0, M + N + 3, y, x] = ey
    th = torch.from_numpy(head)
    outputs = {"anchor_hm": th[:, :M], "part_hm": th[:, M:M + N], "offsets": th[:, M + N:M + N + 2], "embeddings": th[:, M + N + 2:]}
    md = Decoder(args)(outputs, conf_thresh=conf, dist_thresh=dist, return_metadata=True)
    out = {"meta": META, "cfg": np.array([hw * 4, hw * 4, M, N, K, P], np.int64), "head": head, "conf": np.float64(conf),
           "dist": np.float64(dist), "planted_cells": np.array([[1, 30, 30], [M, 40, 30]], np.int64), "l04": l04}
    _store_decode(out, md, args, 1)
    check_margins(np.sort(RU.nms(md["anchor_hm_sig"])[0].numpy().ravel())[::-1][:K + 1], "thresholds anchors")
    check_margins(np.sort(RU.nms(md["part_hm_sig"])[0].numpy().ravel())[::-1][:P + 1], "thresholds parts")
    # the construction must really hit the edges in the reference's own arithmetic
    ann = md["annotation"][0]
    by_xy = {(round(o.x / 4), round(o.y / 4)): o for o in ann.objects}
    assert set(by_xy) == {(0, 10), (30, 30), (50, 50)}, set(by_xy)
    assert by_xy[(30, 30)].anchor.score == float(c32) and by_xy[(30, 30)].parts == []          # emitted, part-less
    assert [round(p.y / 4) for p in by_xy[(0, 10)].parts] == [16]                                # P1 attached, P0 (== thresh) not
    assert len(by_xy[(50, 50)].parts) == 1
    assert any(k.score == float(c32) for k in md["raw_parts"][0])                                # P2 kept in raw_parts
    assert (md["topk_anchor"][0] == -1).sum() == K - 2 and (md["topk_kp"][0] > 0).sum() == 4     # masked scores
    np.savez_compressed(HERE / "decode_thresholds.npz", **out)
    print("thresholds: objects", [(round(o.x / 4), round(o.y / 4), len(o.parts)) for o in ann.objects])


def quantised_head(q_hm, q_reg, cells, vals):
    """(M+N, h, w) int16 logits / 1024, (4, h, w) int8 / 16, planted fp32 cells -> (M+N+4, h, w) fp32 (tests/helpers.py mirrors it)."""
    reg = (q_reg.astype(np.float32) / np.float32(16)).reshape(4, -1)
    reg[:, cells] = vals.T
    return np.concatenate([q_hm.astype(np.float32) / np.float32(1024), reg.reshape(q_reg.shape)], 0)


def gen_evaluate16():
    """BASELINE configs[0]: `evaluate` on 16 synthetic 512x512-input samples stored as PNG + JSON, 2 labels / 1 part,
    anchor_name=stem.  The reference side of the pipeline is run here exactly as cli/evaluate.py:20-45 composes it, minus
    the torchvision image ops (absent; the image content does not influence anything below because the head tensors are
    planted): ImageAnnotation.from_json -> img_size = image size (dataset.py:41-44) -> Resize's annotation.resized
    (transforms.py:58) -> Encode (clips the annotation in place, transforms.py:154) -> Decoder(return_metadata=True) on a
    head synthesised from the encoded targets -> Evaluator.accumulate(prediction, annotation, raw_parts, True, True).
    Stored: the scenes (in ORIGINAL image pixels, incl. out-of-frame keypoints and an empty image), the image sizes, the
    quantised head tensors, and the Evaluator's counters / accuracy lists / CSV."""
    import hashlib
    import tempfile
    M, N, K, P, img = 2, 1, 20, 40, 512
    args = make_args(M, N, K, P)
    args.labels = {"bean": 0, "maize": 1}; args.parts = {"leaf": 0}
    args._r_labels = {0: "bean", 1: "maize"}; args._r_parts = {0: "leaf"}
    args.width = args.height = img; args.dist_threshold = 0.05; args.csi_threshold = 0.75
    ev = Evaluator(args)
    dec_ref, enc_ref = Decoder(args), Encode(args)
    sizes = [(512, 512), (640, 480), (800, 608), (512, 384)]
    out = {"meta": META, "cfg": np.array([img, img, M, N, K, P], np.int64), "noise": np.float64(0.5), "reg_noise": np.float64(0.4)}
    tmp = Path(tempfile.mkdtemp())
    for n in range(16):
        rng = np.random.default_rng(7000 + n)
        iw, ih = sizes[n % 4]
        objs = O.synthetic_scene(rng, iw, ih, M, N, 4, 10, 0, 3)
        if n == 3:
            objs.append((0, -7.5, ih + 3.0, [(0, iw + 12.0, 5.0)]))          # out of frame: clipped by Encode before the Evaluator
        if n == 5:
            objs = []
        js = {"image_path": str(tmp / f"img_{n:02d}.png"), "img_size": [iw, ih],
              "objects": [{"label": args._r_labels[l], "box": None,
                           "parts": [{"kind": "stem", "location": {"x": x, "y": y}}]
                           + [{"kind": args._r_parts[k], "location": {"x": px, "y": py}} for (k, px, py) in ps]}
                          for (l, x, y, ps) in objs]}
        f = tmp / f"img_{n:02d}.json"
        f.write_text(json.dumps(js))
        ann = RU.ImageAnnotation.from_json(f, "stem")
        ann.img_size = (iw, ih)                                                # dataset.py:43-44 (PIL image.size)
        resized = ann.resized((iw, ih), (img, img))                            # transforms.py:58
        e = enc_ref(torch.zeros(3, img, img), resized)                         # clips `resized` in place
        enp = {k: (v.numpy() if isinstance(v, torch.Tensor) else v) for k, v in e.items()}
        for attempt in range(20):                                              # re-draw the head noise until the ranking is tie-free
            seed = 9000 + n + 100 * attempt
            raw = O.head_from_targets(np.random.default_rng(seed), enp, M, N, noise=0.5, reg_noise=0.4)
            # The head travels in the fixture itself (exp / log differ by an ulp between host CPUs, so it cannot be rebuilt
            # bit-exactly elsewhere): heatmap logits on a 1/1024 grid (int16), regression channels on a 1/16 grid (int8), the
            # planted offset / embedding cells in full fp32.  What the reference decodes below is exactly this quantised head.
            q_hm = np.round(raw[:M + N] * 1024).astype(np.int16)
            q_reg = np.clip(np.round(raw[M + N:] * 16), -127, 127).astype(np.int8)
            cells = np.unique(np.concatenate([enp["anchor_inds"][enp["anchor_mask"]], enp["part_inds"][enp["part_mask"]]])).astype(np.int32)
            vals = raw[M + N:].reshape(4, -1)[:, cells].T.copy()
            head = quantised_head(q_hm, q_reg, cells, vals)
            th = torch.from_numpy(head)[None]
            md = dec_ref({"anchor_hm": th[:, :M], "part_hm": th[:, M:M + N], "offsets": th[:, M + N:M + N + 2],
                          "embeddings": th[:, M + N + 2:]}, return_metadata=True)
            try:
                for grp, kk, sig in (("anchors", K, md["anchor_hm_sig"]), ("parts", P, md["part_hm_sig"])):
                    top = np.sort(RU.nms(sig)[0].numpy().ravel())[::-1][:kk + 1]
                    live = top[top > 0.5 - 1e-3]                                   # slots below the threshold never reach an output
                    check_margins(live, f"evaluate16 {grp} img{n}")
                    assert (np.abs(top - 0.5) > 1e-5).all(), f"img{n}: a {grp} score sits on the confidence threshold"
                break
            except AssertionError as err:
                print("  re-drawing head", n, "->", err)
        else:
            raise AssertionError(f"no tie-free head for image {n}")
        out[f"head{n}_hm_q1024"] = q_hm; out[f"head{n}_reg_q16"] = q_reg
        out[f"head{n}_cells"] = cells; out[f"head{n}_cell_vals"] = vals
        ev.accumulate(md["annotation"][0], e["annotation"], md["raw_parts"][0], True, True)
        so, sp = flat_scene(objs)
        out[f"scene{n}_objs"] = so; out[f"scene{n}_parts"] = sp; out[f"size{n}"] = np.array([iw, ih], np.int64)
        out[f"n_pred{n}"] = np.array([len(md["annotation"][0].objects), len(md["raw_parts"][0])], np.int64)
    for sec, evals in (("anchor", ev.anchor_eval), ("part", ev.part_eval), ("csi", ev.csi_eval), ("classif", ev.classification_eval)):
        out[f"{sec}_labels"] = np.array(list(evals.labels))
        out[f"{sec}_counts"] = np.array([[e.tp, e.npos, e.ndet] for _, e in evals.items()], np.int64)
        for label, e in evals.items():
            out[f"{sec}_acc_{label}"] = np.array(e.acc, np.float64)
    out["csv"] = np.array(ev._csv_kps_str())
    np.savez_compressed(HERE / "evaluate16.npz", **out)
    print("evaluate16:", ev.anchor_eval.reduce(), "|", ev.part_eval.reduce(), "|", ev.csi_eval.reduce())


def gen_annotation_transforms(rng):
    """Annotation side of the augmentations (utils.py:364-415: clip, hflip, vflip incl. boxes) and the colour map (utils.py:476-479)."""
    out = {"meta": META}
    objs = []
    for i in range(5):
        x, y = float(rng.uniform(-5, 210)), float(rng.uniform(-5, 110))
        box = RU.Box(x - 10.5, y - 7.25, x + 12.0, y + 9.5) if i % 2 == 0 else None
        kps = [RU.Keypoint("leaf", float(rng.uniform(0, 200)), float(rng.uniform(0, 100))) for _ in range(i % 3)]
        objs.append(RU.Object("bean" if i % 2 else "maize", RU.Keypoint("stem", x, y), kps, box))
    ann = RU.ImageAnnotation("a.jpg", objs)

    def flat(a):
        rows = []
        for o in a.objects:
            b = o.box
            rows.append([o.x, o.y] + ([b.x_min, b.y_min, b.x_max, b.y_max] if b is not None else [np.nan] * 4) + [len(o.parts)])
            rows += [[k.x, k.y, np.nan, np.nan, np.nan, np.nan, -1] for k in o.parts]
        return np.array(rows, np.float64)

    import copy
    out["input"] = flat(ann)
    out["hflip"] = flat(RU.hflip_annotation(copy.deepcopy(ann), (200, 100)))
    out["vflip"] = flat(RU.vflip_annotation(copy.deepcopy(ann), (200, 100)))
    out["hvflip"] = flat(RU.vflip_annotation(RU.hflip_annotation(copy.deepcopy(ann), (200, 100)), (200, 100)))
    out["clip"] = flat(RU.clip_annotation(copy.deepcopy(ann), (200, 100)))
    out["resized"] = flat(copy.deepcopy(ann).resize((200, 100), (512, 384)))
    names = ["bean", "maize", "leaf", "stem", "haricot"]
    out["color_names"] = np.array(names)
    out["colors"] = np.array([RU.get_unique_color_map(names)[n] for n in names], np.int64)
    np.savez_compressed(HERE / "annotation_transforms.npz", **out)


def gen_fpn_head(rng):
    torch.manual_seed(1234)
    fpn = Fpn(16, 8).train(); head = Head(8, 7)
    x = torch.randn(2, 8, 6, 10); sc = torch.randn(2, 16, 12, 20)
    y = fpn(x, sc)
    out = {"meta": META, "x": x.numpy(), "shortcut": sc.numpy(), "fpn_out": y.detach().numpy(),
           "head_out": head(y).detach().numpy()}
    for k, v in fpn.state_dict().items():
        out[f"fpn.{k}"] = v.numpy()
    for k, v in head.state_dict().items():
        out[f"head.{k}"] = v.numpy()
    np.savez_compressed(HERE / "fpn_head.npz", **out)


def gen_small_utils(rng):
    """`draw_heatmaps` (visualization.py:53-91: colour of the arg-max channel times its value, truncated to uint8), `AverageMeter`
    (utils.py:311-324) and `dict_grouping` (utils.py:470-474).  The other drawing helpers of visualization.py go through torchvision's
    `to_pil_image`, which is absent here: they are pinned by their PIL primitives in tests/test_host_cpu.py instead."""
    from sdnet.utils.visualization import draw_heatmaps
    args = make_args(3, 2, 8, 8)
    args._label_color_map = RU.get_unique_color_map(args.labels)
    args._part_color_map = RU.get_unique_color_map(args.parts)
    a = torch.from_numpy(rng.random((3, 12, 16)).astype(np.float32))
    q = torch.from_numpy(rng.random((2, 12, 16)).astype(np.float32))
    a[:, 0, 0] = torch.tensor([0.25, 0.25, 0.1])                      # a tie: the first maximal channel gives the colour
    ca, cq = draw_heatmaps(a, q, args)
    m = RU.AverageMeter()
    vals = rng.random(7)
    avgs = [m.update(float(v)) for v in vals]
    words = ["pear", "fig", "plum", "kiwi", "lime", "date", "peach"]
    grouped = RU.dict_grouping(words, key=len)
    out = {"anchor_hm": a.numpy(), "part_hm": q.numpy(), "anchor_rgb": ca.numpy(), "part_rgb": cq.numpy(),
           "label_colors": np.array([args._label_color_map[args._r_labels[i]] for i in range(3)], dtype=np.int64),
           "part_colors": np.array([args._part_color_map[args._r_parts[i]] for i in range(2)], dtype=np.int64),
           "meter_values": vals, "meter_avgs": np.array(avgs), "meter_sum": np.array(m.sum), "meter_count": np.array(m.count),
           "words": np.array(words), "grouped": np.array(json.dumps({str(k): v for k, v in grouped.items()})), "meta": np.array(META)}
    np.savez_compressed(HERE / "small_utils.npz", **out)


def hf_to_oracle_key(k):
    """HF `ResNetModel` state_dict key -> the key of the same tensor in the reference's `Network` (`network.py:43-50`: `adpater` =
    conv1 / bn1 / relu / maxpool, `down1..4` = layer1..4 of torchvision's resnet34)."""
    part = {"convolution": None, "normalization": None}
    if k.startswith("embedder.embedder."):
        kind, leaf = k[len("embedder.embedder."):].split(".", 1)
        return f"adpater.{0 if kind == 'convolution' else 1}.{leaf}"
    s, l, rest = re.match(r"encoder\.stages\.(\d+)\.layers\.(\d+)\.(.*)", k).groups()
    base = f"down{int(s) + 1}.{l}."
    if rest.startswith("shortcut."):
        kind, leaf = rest[len("shortcut."):].split(".", 1)
        return base + f"downsample.{0 if kind == 'convolution' else 1}.{leaf}"
    i, kind, leaf = re.match(r"layer\.(\d)\.(\w+)\.(.*)", rest).groups()
    return base + (f"conv{int(i) + 1}." if kind == "convolution" else f"bn{int(i) + 1}.") + leaf


def gen_resnet34_second_source():
    """An INDEPENDENT published port of torchvision's ResNet-34 as the second source for the trunk the reference takes from torchvision
    (`network.py:3,41,43-50`; torchvision itself is absent from this container): Hugging Face `transformers`' `ResNetModel` with
    `layer_type="basic"`, depths [3,4,6,3], widths [64,128,256,512] -- 21 284 672 parameters, torchvision's count without the fc layer.
    Weights come from `oracle.build_reference_network(seed)` (so the test can rebuild them from the seed) plus non-trivial running
    statistics, loaded INTO the HF model through `hf_to_oracle_key`; the fixture holds what the HF model computes."""
    import hashlib
    stubs = {k: sys.modules.pop(k) for k in list(sys.modules) if k.split(".")[0] == "torchvision"}   # transformers probes for torchvision
    try:
        import transformers
        from transformers import ResNetConfig, ResNetModel
    finally:
        sys.modules.update(stubs)
    seed = 34
    onet = O.build_reference_network(2, 1, seed=seed)
    osd = onet.state_dict()
    g = torch.Generator().manual_seed(seed + 1)
    for k in sorted(osd):
        if k.endswith("running_mean"):
            osd[k].copy_(0.2 * torch.randn(osd[k].shape, generator=g))
        elif k.endswith("running_var"):
            osd[k].copy_(0.5 + torch.rand(osd[k].shape, generator=g))
    hf = ResNetModel(ResNetConfig(layer_type="basic", depths=[3, 4, 6, 3], hidden_sizes=[64, 128, 256, 512], embedding_size=64))
    assert sum(p.numel() for p in hf.parameters()) == 21_284_672
    hsd = hf.state_dict()
    mapped = {k: osd[hf_to_oracle_key(k)].clone() for k in hsd}
    trunk = {k for k in osd if k.startswith(("adpater.", "down"))}
    assert {hf_to_oracle_key(k) for k in hsd} == trunk, "the HF trunk and the reference's trunk keys are not in bijection"
    assert all(hsd[k].shape == mapped[k].shape for k in hsd)
    hf.load_state_dict(mapped, strict=True)
    h = hashlib.sha256()
    for k in sorted(trunk):
        h.update(k.encode()); h.update(osd[k].numpy().tobytes())
    x = torch.randn(2, 3, 96, 128, generator=torch.Generator().manual_seed(seed + 2))
    out = {"meta": np.array(json.dumps({"torch": torch.__version__, "transformers": transformers.__version__, "seed": seed,
                                        "source": "transformers.ResNetModel(layer_type=basic, depths=[3,4,6,3])"})),
           "seed": np.array(seed), "x": x.numpy(), "trunk_sha256": np.array(h.hexdigest())}
    hf.eval()
    with torch.no_grad():
        hs = hf(x, output_hidden_states=True).hidden_states
    assert len(hs) == 5
    for i, t in enumerate(hs):
        out[f"eval_stage{i}"] = t.numpy()                              # 0 = stem + max-pool, 1..4 = layer1..4
    hf.train()
    with torch.no_grad():
        hs = hf(x, output_hidden_states=True).hidden_states
    for i, t in enumerate(hs):
        out[f"train_stage{i}"] = t.numpy()
    for k, v in hf.state_dict().items():                               # running statistics after ONE training-mode forward
        if k.endswith(("running_mean", "running_var", "num_batches_tracked")):
            out["after." + hf_to_oracle_key(k)] = v.numpy()
    np.savez_compressed(HERE / "resnet34_second_source.npz", **out)


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "small_utils":          # one new fixture without touching the others
        gen_small_utils(np.random.default_rng(5))
        print("small_utils.npz written to", HERE)
        raise SystemExit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "resnet34_second_source":
        gen_resnet34_second_source()
        print("resnet34_second_source.npz written to", HERE)
        raise SystemExit(0)
    rng = np.random.default_rng(20261003)
    gen_prims(rng)
    gen_encode_decode(rng, "scene_cfg512", 512, 2, 1, 20, 40, n_img=3, n_min=6, n_max=12, noise=0.05)
    gen_encode_decode(rng, "scene_small256", 256, 3, 2, 12, 24, n_img=4, n_min=3, n_max=8, noise=0.2)
    gen_truncation(rng)
    gen_fpn_head(rng)
    gen_evaluator(np.random.default_rng(77))
    gen_thresholds(np.random.default_rng(404))
    gen_annotation_transforms(np.random.default_rng(31))
    gen_evaluate16()
    gen_small_utils(np.random.default_rng(5))
    gen_resnet34_second_source()
    print("goldens written to", HERE)
