"""Evaluator against the reference's own Evaluator on decoded noisy scenes (tests/golden/evaluator.npz).
Host-side code: CPU only."""
from argparse import Namespace

import numpy as np
import pytest

from tests.helpers import scene_from_flat


def build(golden_dir, keep=None):
    from structuredetector_amd.model import Evaluator
    from structuredetector_amd.utils import ImageAnnotation, Keypoint, Object
    g = np.load(golden_dir / "evaluator.npz")
    labels, parts = {"bean": 0, "maize": 1}, {"leaf": 0}
    rl, rp = {0: "bean", 1: "maize"}, {0: "leaf"}
    args = Namespace(labels=labels, parts=parts, width=512, height=512, dist_threshold=0.05, csi_threshold=0.75)
    ev = Evaluator(args)
    n = 0
    while f"gt{n}_objs" in g:
        gt = ImageAnnotation(f"g{n}", [Object(rl[l], Keypoint("stem", x, y), [Keypoint(rp[k], px, py) for k, px, py in ps])
                                       for l, x, y, ps in scene_from_flat(g[f"gt{n}_objs"], g[f"gt{n}_parts"])],
                             img_size=tuple(int(v) for v in g[f"gt{n}_size"]))
        po, pp = g[f"pred{n}_objs"], g[f"pred{n}_parts"]
        objs = []
        for oi, (l, x, y, s) in enumerate(po):
            kps = [Keypoint(rp[int(k)], px, py, ps) for (o, k, px, py, ps) in pp if int(o) == oi]
            objs.append(Object(rl[int(l)], Keypoint("stem", x, y, s), kps))
        pred = ImageAnnotation(f"p{n}", objs)
        raw = [Keypoint(rp[int(k)], x, y, s) for (k, x, y, s) in g[f"raw{n}"]]
        if keep is None or keep(n):
            ev.accumulate(pred, gt, raw, True, True)
        n += 1
    assert n == 6
    return g, ev


def test_evaluator_matches_reference(golden_dir):
    g, ev = build(golden_dir)
    for sec, evals in (("anchor", ev.anchor_eval), ("part", ev.part_eval), ("csi", ev.csi_eval), ("classif", ev.classification_eval)):
        assert list(evals.labels) == list(g[f"{sec}_labels"])
        counts = np.array([[e.tp, e.npos, e.ndet] for _, e in evals.items()], np.int64)
        np.testing.assert_array_equal(counts, g[f"{sec}_counts"], err_msg=sec)
        for label, e in evals.items():
            np.testing.assert_array_equal(np.array(e.acc, np.float64), g[f"{sec}_acc_{label}"], err_msg=f"{sec} {label}")
    assert ev._csv_kps_str() == str(g["csv"])
    tot = ev.anchor_eval.reduce()
    assert tot.tp == 38 and tot.ndet == 65 and 0 < ev.csi_eval.reduce().tp < tot.tp      # the golden really has FPs and CSI misses
    assert "Anchor Location" in repr(ev)
    ev.pretty_print()


def test_evaluator_merge_is_associative(golden_dir):
    """Data-parallel evaluation: summing per-rank Evaluators equals one Evaluator over all images."""
    from structuredetector_amd.model import Evaluation, Evaluations
    g, ev = build(golden_dir)
    _, ev2 = build(golden_dir)
    ev2.merge(ev)
    for a, b in ((ev.anchor_eval, ev2.anchor_eval), (ev.csi_eval, ev2.csi_eval)):
        for label in a.labels:
            assert b[label].tp == 2 * a[label].tp and b[label].npos == 2 * a[label].npos and len(b[label].acc) == 2 * len(a[label].acc)
    e = Evaluation()
    assert e.precision == 1 and e.recall == 1 and e.f1_score == 1 and e.csi == 1 and np.isnan(e.avg_acc)
    assert Evaluation(0, 0, 3).precision == 0 and Evaluation(0, 3, 0).recall == 0
    u = Evaluations(["a"]) | Evaluations(["b"])
    assert set(u.labels) == {"a", "b"}


def test_evaluator_merge_over_8_shards_equals_single_evaluator(golden_dir):
    """World = 8 rehearsal of data-parallel evaluation: the images sharded [rank::8] (two ranks get none), one Evaluator per rank,
    merged on rank 0 == one Evaluator over all images (counters exactly, accuracy lists as multisets, the same metrics)."""
    _, whole = build(golden_dir)
    shards = [build(golden_dir, keep=lambda n, r=r: n % 8 == r)[1] for r in range(8)]
    merged = shards[0]
    for other in shards[1:]:
        merged.merge(other)
    for a, b in ((whole.anchor_eval, merged.anchor_eval), (whole.part_eval, merged.part_eval), (whole.csi_eval, merged.csi_eval),
                 (whole.classification_eval, merged.classification_eval)):
        assert list(a.labels) == list(b.labels)
        for label in a.labels:
            assert (a[label].tp, a[label].npos, a[label].ndet) == (b[label].tp, b[label].npos, b[label].ndet), label
            assert sorted(a[label].acc) == sorted(b[label].acc), label
        assert a.reduce().f1_score == b.reduce().f1_score
    assert whole._csv_kps_str() == merged._csv_kps_str()


def test_evaluate16_directory_reader_and_evaluator_vs_reference(golden_dir, tmp_path):
    """BASELINE configs[0] without the GPU: PNG + JSON on disk -> product CropDataset (resize + clip) -> ORACLE decode of
    the planted heads -> product Evaluator == the reference's pipeline (tests/golden/evaluate16.npz: reference
    from_json / Resize / Encode-clip / Decoder / Evaluator).  Pins the reader, the clip and the oracle's
    decode + assembly on 16 more images; the GPU test runs the same directory through `evaluate` with the HIP decoder."""
    from oracle import sdnet_oracle as O
    from structuredetector_amd.data import CropDataset
    from structuredetector_amd.model import Evaluator
    from structuredetector_amd.utils import ImageAnnotation, Keypoint, Object
    from tests.helpers import EVAL16_LABELS, EVAL16_PARTS, assert_evaluator_equals_golden, write_evaluate16_dir
    g = np.load(golden_dir / "evaluate16.npz")
    heads = write_evaluate16_dir(g, tmp_path / "valid")
    W, H, M, N, K, P = (int(v) for v in g["cfg"])
    rl, rp = {v: k for k, v in EVAL16_LABELS.items()}, {v: k for k, v in EVAL16_PARTS.items()}
    args = Namespace(labels=EVAL16_LABELS, parts=EVAL16_PARTS, width=W, height=H, dist_threshold=0.05, csi_threshold=0.75,
                     anchor_name="stem")
    ds = CropDataset(args, tmp_path / "valid")
    assert len(ds) == 16
    ev = Evaluator(args)
    for n in range(16):
        image, ann = ds[n]
        assert tuple(image.shape) == (3, H, W) and tuple(ann.img_size) == tuple(int(v) for v in g[f"size{n}"])
        assert all(0 <= o.x <= W - 1 and 0 <= o.y <= H - 1 for o in ann.objects)           # clipped like Encode does
        h = heads[n][None]
        t = O.decode_tensors(h[:, :M], h[:, M:M + N], h[:, M + N:M + N + 2], h[:, M + N + 2:], K, P, 0.5, 0.1)
        objs = O.assemble_objects(t, 0, 0.5, 4.0, W // 4, H // 4)
        pred = ImageAnnotation("batch_0", [Object(rl[l], Keypoint("stem", *a), [Keypoint(rp[k], x, y, s) for (k, x, y, s) in ps])
                                           for (l, a, ps) in objs])
        raw = [Keypoint(rp[k], x, y, s) for (k, x, y, s) in O.raw_parts(t, 0, 0.5, 4.0, W // 4, H // 4)]
        assert [len(pred.objects), len(raw)] == list(g[f"n_pred{n}"])
        ev.accumulate(pred, ann, raw, True, True)
    assert_evaluator_equals_golden(ev, g)
    assert ev.anchor_eval.reduce().ndet == 125 and ev.csi_eval.reduce().tp == 88


def test_prefetching_reader_yields_the_sequential_items_in_order(golden_dir, tmp_path):
    """data/feeder.prefetch_items (the reader behind `evaluate`, the validation pass and `detect`): the items of the sequential walk,
    bit for bit and in order, whatever the pool size / look-ahead; a reader error surfaces at its position; an early stop leaves no
    thread behind."""
    import threading

    import torch
    from structuredetector_amd.data import CropDataset
    from structuredetector_amd.data.feeder import prefetch_items
    from tests.helpers import EVAL16_LABELS, EVAL16_PARTS, write_evaluate16_dir
    g = np.load(golden_dir / "evaluate16.npz")
    write_evaluate16_dir(g, tmp_path / "valid")
    W, H = (int(v) for v in g["cfg"][:2])
    args = Namespace(labels=EVAL16_LABELS, parts=EVAL16_PARTS, width=W, height=H, anchor_name="stem")
    ds = CropDataset(args, tmp_path / "valid")
    want = [ds[i] for i in range(len(ds))]
    for workers, depth in ((1, 1), (3, 2), (8, 32)):
        got = list(prefetch_items(ds, workers, depth))
        assert len(got) == len(want)
        for (im, an), (im0, an0) in zip(got, want):
            assert torch.equal(im, im0) and an.image_name == an0.image_name and tuple(an.img_size) == tuple(an0.img_size)
            assert [(o.name, o.x, o.y, len(o.parts)) for o in an.objects] == [(o.name, o.x, o.y, len(o.parts)) for o in an0.objects]

    class Broken:
        def __len__(self):
            return 6

        def __getitem__(self, i):
            if i == 3:
                raise ValueError("unreadable sample 3")
            return i
    it = prefetch_items(Broken(), 2, 4)
    assert [next(it), next(it), next(it)] == [0, 1, 2]
    with pytest.raises(ValueError, match="unreadable sample 3"):
        next(it)
    it = prefetch_items(ds, 4, 8)
    next(it); it.close()                                     # early stop: the pool is shut down by the generator's finally
    assert not [t for t in threading.enumerate() if t.name.startswith("sd-read")]


def test_fast_accumulate_equals_the_per_metric_statement_on_adversarial_scenes():
    """`Evaluator.accumulate` computes all four metrics of an image from three distance matrices; the public `eval_anchor / eval_part /
    eval_csi / eval_classif / compute_csi` methods (the reference's API, evaluator.py:244-474,538-581) stay as the statement it must
    equal: identical counters AND accuracy lists (same values, same order) on scenes with coincident keypoints, tied scores, tied
    distances, empty sides, several part kinds, and predictions on / next to the threshold."""
    from structuredetector_amd.model import Evaluator
    from structuredetector_amd.utils import ImageAnnotation, Keypoint, Object
    labels, parts = {"bean": 0, "maize": 1}, {"leaf": 0, "pod": 1}
    args = Namespace(labels=labels, parts=parts, width=512, height=384, dist_threshold=0.05, csi_threshold=0.75)
    rng = np.random.default_rng(7)
    fast, slow = Evaluator(args), Evaluator(args)
    grid = lambda n: rng.integers(0, 12, n) * 16.0                      # a coarse grid: coincident points and exactly tied distances
    for n in range(300):
        size = [(640, 480), (512, 384), (1000, 300)][n % 3]
        n_gt, n_pr = int(rng.integers(0, 7)), int(rng.integers(0, 8))
        gts = [Object(["bean", "maize"][int(rng.integers(2))], Keypoint("stem", float(x), float(y)),
                      [Keypoint(["leaf", "pod"][int(rng.integers(2))], float(x + dx), float(y + dy)) for dx, dy in rng.integers(-2, 3, (int(rng.integers(0, 4)), 2)) * 8.0])
               for x, y in zip(grid(n_gt), grid(n_gt))]
        prs = [Object(["bean", "maize"][int(rng.integers(2))], Keypoint("stem", float(x), float(y), float(rng.integers(5, 10)) / 10),
                      [Keypoint(["leaf", "pod"][int(rng.integers(2))], float(x + dx), float(y + dy), float(rng.integers(5, 10)) / 10)
                       for dx, dy in rng.integers(-2, 3, (int(rng.integers(0, 4)), 2)) * 8.0])
               for x, y in zip(grid(n_pr) + rng.choice([0.0, 0.0, 3.0, 19.2, 24.0], n_pr), grid(n_pr))]
        raw = [Keypoint(kp.kind, kp.x, kp.y, kp.score) for o in prs for kp in o.parts] + \
              [Keypoint("leaf", float(x), float(y), 0.5) for x, y in zip(grid(2), grid(2))]
        gt, pred = ImageAnnotation(f"g{n}", gts, img_size=size), ImageAnnotation(f"p{n}", prs)
        fast.accumulate(pred, gt, raw if n % 5 else None, eval_csi=bool(n % 2), eval_classif=bool(n % 3))
        slow._accumulate_by_metric(pred, gt, raw if n % 5 else None, eval_csi=bool(n % 2), eval_classif=bool(n % 3))
    for name in ("anchor_eval", "part_eval", "csi_eval", "classification_eval"):
        for (label, a), (_, b) in zip(getattr(fast, name).items(), getattr(slow, name).items()):
            assert (a.tp, a.npos, a.ndet) == (b.tp, b.npos, b.ndet), (name, label)
            assert a.acc == b.acc, (name, label)
    tot = fast.anchor_eval.reduce()
    assert 0 < tot.tp < tot.ndet and 0 < fast.csi_eval.reduce().tp and 0 < fast.classification_eval.reduce().tp < tot.tp

    class Custom(Evaluator):                                              # a subclass that overrides a metric keeps its override in charge
        def compute_csi(prediction, target, dist_thresh):
            return 1.0
    c = Custom(args)
    c.accumulate(pred, gt, None, eval_csi=True)
