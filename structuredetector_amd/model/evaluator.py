"""`Evaluator`: keypoint / structure metrics of the `evaluate` entry point (the first "next" row).

Same metrics, thresholds and table layout as the reference's `Evaluator` / `Evaluations` / `Evaluation`
(src/sdnet/model/evaluator.py:13-646): greedy matching of predictions (by descending score) to their NEAREST
ground truth -- a prediction whose nearest ground truth is already taken is a false positive, there is no
second choice (:244-286) -- for anchors (:244), raw parts (:288), whole structures by CSI (:380, :538) and the
count-classification metric (:427).  The O(preds x gts) Python loops of the reference become one distance matrix
and an argmin per image (numpy, float64 like `np.hypot` in `Keypoint.distance`), which keeps `evaluate` from being
dominated by host matching once the decoder takes microseconds.  Counters merge associatively (`+=`), so ranks of a
data-parallel evaluation can be combined by summing (`Evaluator.merge`).
"""
from __future__ import annotations

from collections import defaultdict
from pathlib import Path

import numpy as np


def greedy_nearest(pred_xy, pred_score, gt_xy, threshold, inclusive=False):
    """Match predictions (score-descending, stable) to their nearest ground truth.
    Returns (number of true positives, distances of the true positives in match order)."""
    if len(pred_xy) == 0 or len(gt_xy) == 0:
        return 0, []
    p = np.asarray(pred_xy, np.float64).reshape(-1, 2)
    g = np.asarray(gt_xy, np.float64).reshape(-1, 2)
    order = np.argsort(-np.asarray(pred_score, np.float64), kind="stable")
    d = np.hypot(p[order, None, 0] - g[None, :, 0], p[order, None, 1] - g[None, :, 1])
    j = d.argmin(axis=1)                                   # first minimum, like the reference's strict `<` scan
    dmin = d[np.arange(len(order)), j]
    ok = (dmin <= threshold) if inclusive else (dmin < threshold)
    rows = np.nonzero(ok)[0]
    _, first = np.unique(j[rows], return_index=True)       # the first (best-scored) claimant of each ground truth wins
    winners = np.sort(rows[first])
    return len(winners), dmin[winners].tolist()


def _dist_rows(px, py, gx, gy):
    """Distance matrix (predictions x ground truths) as nested lists of Python floats; `np.hypot` like `Keypoint.distance` (utils.py:31-32)."""
    if not px or not gx:
        return []
    return np.hypot(np.subtract.outer(np.asarray(px, np.float64), np.asarray(gx, np.float64)),
                    np.subtract.outer(np.asarray(py, np.float64), np.asarray(gy, np.float64))).tolist()


def _greedy_rows(rows, preds, gts, threshold, inclusive):
    """`greedy_nearest` on rows of a precomputed distance matrix: `preds` = prediction indices in score-descending order, `gts` = ground-truth
    indices; a prediction takes its NEAREST ground truth (first minimum: strict `<` scan) if that is within the threshold and still free."""
    if not preds or not gts:
        return 0, ()
    taken, dists = set(), []
    for p in preds:
        row = rows[p]
        jm, dm = None, None
        for j in gts:
            d = row[j]
            if dm is None or d < dm:
                dm, jm = d, j
        if (dm <= threshold if inclusive else dm < threshold) and jm not in taken:
            taken.add(jm)
            dists.append(dm)
    return len(dists), dists


class Evaluation:
    def __init__(self, tp=0, npos=0, ndet=0, acc=None, counts=None):
        assert tp >= 0 and ndet >= 0 and npos >= 0, "tp, npos and ndet should be positive"
        assert tp <= ndet, "tp must be lower than or equal to ndet"
        assert tp <= npos, "tp must be lower than or equal to npos"
        self.tp, self.npos, self.ndet = tp, npos, ndet
        self.acc = acc or []
        self.count_errors = counts or []

    def reset(self):
        self.__init__()

    def __iadd__(self, other):
        self.tp += other.tp; self.npos += other.npos; self.ndet += other.ndet
        if other.acc:
            self.acc.extend(other.acc)                      # in place: `self.acc + other.acc` re-copied the whole history for every image
        if other.count_errors:
            self.count_errors.extend(other.count_errors)
        return self

    def __add__(self, other):
        out = Evaluation(self.tp, self.npos, self.ndet, list(self.acc), list(self.count_errors))
        out += other
        return out

    fp = property(lambda s: s.ndet - s.tp)
    fn = property(lambda s: s.npos - s.tp)

    @property
    def csi(self):
        den = self.npos + self.ndet - self.tp
        return self.tp / den if den != 0 else 1

    @property
    def precision(self):
        return self.tp / self.ndet if self.ndet != 0 else 1 if self.npos == 0 else 0

    @property
    def recall(self):
        return self.tp / self.npos if self.npos != 0 else 1 if self.ndet == 0 else 0

    @property
    def f1_score(self):
        s = self.npos + self.ndet
        return 2 * self.tp / s if s != 0 else 1

    @property
    def avg_acc(self):
        return np.mean(self.acc) if len(self.acc) != 0 else float("nan")

    @property
    def acc_err(self):
        return np.std(self.acc) / np.sqrt(len(self.acc)) if len(self.acc) != 0 else float("nan")

    HEADERS = ("Gts.", "Preds.", "Rec.", "Prec.", "F1 Score", "L. Acc.", "L. Err.")

    def stats(self):
        return (f"{self.npos}", f"{self.ndet}", f"{self.recall:.2%}", f"{self.precision:.2%}", f"{self.f1_score:.2%}",
                f"{self.avg_acc:.4%}", f"{self.acc_err:.4%}")

    def __repr__(self):
        return (f"f1: {self.f1_score:.2%}, rec: {self.recall:.2%}, prec: {self.precision:.2%}, npos: {self.npos}, ndet: {self.ndet}, "
                f"tp/fp/fn: {self.tp}/{self.fp}/{self.fn}, avg_acc: {self.avg_acc:.2}")


class Evaluations:
    def __init__(self, labels=None):
        self.evals = {label: Evaluation() for label in labels} if labels else {}

    def reset(self):
        for e in self.evals.values():
            e.reset()

    labels = property(lambda s: s.evals.keys())

    def items(self):
        return self.evals.items()

    def __getitem__(self, label):
        return self.evals[label]

    def __setitem__(self, label, item):
        self.evals[label] = item

    def __len__(self):
        return len(self.evals)

    def __iadd__(self, other):
        assert self.labels == other.labels, "The Evaluations should have the same labels"
        for label, e in other.items():
            self.evals[label] += e
        return self

    def __add__(self, other):
        assert self.labels == other.labels, "The Evaluations should have the same labels"
        out = Evaluations()
        out.evals = {label: self.evals[label] + e for label, e in other.items()}
        return out

    def __or__(self, other):
        out = Evaluations()
        out.evals = {label: self[label] + other[label] for label in self.labels & other.labels}
        out.evals.update({label: self[label] for label in self.labels - other.labels})
        out.evals.update({label: other[label] for label in other.labels - self.labels})
        return out

    def reduce(self):
        total = Evaluation()
        for e in self.evals.values():
            total += e
        return total

    def __repr__(self):
        lines = [f"total: {self.reduce()}"] if len(self) > 1 else []
        return "\n".join(lines + [f"{label}: {e}" for label, e in self.items()])


def _by(items, key):
    out = defaultdict(list)
    for it in items:
        out[key(it)].append(it)
    return out


class Evaluator:
    def __init__(self, args):
        self.args = args
        self.labels = args.labels.keys()
        self.kp_labels = args.parts.keys()
        self.reset()

    def reset(self):
        self.anchor_eval = Evaluations(self.labels)
        self.part_eval = Evaluations(self.kp_labels)
        self.csi_eval = Evaluations(self.labels)
        self.classification_eval = Evaluations(Evaluator.get_classification_labels())

    @property
    def kps_eval(self):
        return self.anchor_eval | self.part_eval

    @staticmethod
    def get_classification_labels():
        """Hard-coded in the reference too (evaluator.py:421-425)."""
        return [f"bean_{i}" for i in range(10)] + [f"maize_{i}" for i in range(10)]

    def merge(self, other):
        """Combine the counters of another rank's Evaluator (data-parallel evaluation)."""
        self.anchor_eval += other.anchor_eval; self.part_eval += other.part_eval
        self.csi_eval += other.csi_eval; self.classification_eval += other.classification_eval
        return self

    # ------------------------------------------------------------------ per-image accumulation (evaluator.py:226-242)
    def accumulate(self, prediction, annotation, part_heatmap=None, eval_csi=False, eval_classif=False):
        """evaluator.py:226-242.  All four metrics of one image from THREE distance matrices (anchors x anchors, object parts x object
        parts, raw parts x ground-truth parts; `np.hypot` like `Keypoint.distance`, utils.py:31-32) and plain-Python greedy scans over
        their rows: the per-pair numpy calls of the public `eval_*` / `compute_csi` methods (kept below: the reference's API, and the
        statement the fast path is tested against) cost ~10 us each on 1-3 element arrays, ~50 structure pairs per image."""
        # the fast path restates the six methods below; a subclass (or a patched instance) that overrides any of them gets the
        # reference's own definition of accumulate -- calls to exactly those methods (evaluator.py:226-242)
        names = ("_match_objects", "compute_csi", "eval_anchor", "eval_part", "eval_csi", "eval_classif")
        if all(getattr(type(self), n) is getattr(Evaluator, n) and n not in self.__dict__ for n in names):
            return self._accumulate_fast(prediction, annotation, part_heatmap, eval_csi, eval_classif)
        return self._accumulate_by_metric(prediction, annotation, part_heatmap, eval_csi, eval_classif)

    def accumulate_batch(self, predictions, annotations, part_heatmaps=None, eval_csi=False, eval_classif=False):
        """`accumulate` over the images of one decoded batch, in order (counters and accuracy lists are order-dependent only through the
        list order, which is kept)."""
        part_heatmaps = part_heatmaps if part_heatmaps is not None else [None] * len(predictions)
        for pred, ann, raw in zip(predictions, annotations, part_heatmaps):
            self.accumulate(pred, ann, raw, eval_csi, eval_classif)

    def _accumulate_fast(self, prediction, annotation, part_heatmap, eval_csi, eval_classif):
        a = self.args
        img_size = annotation.img_size
        fx, fy = img_size[0] / a.width, img_size[1] / a.height          # ann.resized((width, height), img_size): x *= ow / iw
        side = min(img_size)
        thr = side * a.dist_threshold

        def flat(objects):
            ax, ay, names, scores, nparts, owner, kx, ky, kinds, kscores = [], [], [], [], [], [], [], [], [], []
            for i, o in enumerate(objects):
                an = o.anchor
                ax.append(an.x * fx); ay.append(an.y * fy); names.append(o.name); scores.append(an.score); nparts.append(len(o.parts))
                for kp in o.parts:
                    owner.append(i); kx.append(kp.x * fx); ky.append(kp.y * fy); kinds.append(kp.kind); kscores.append(kp.score)
            return ax, ay, names, scores, nparts, owner, kx, ky, kinds, kscores

        pax, pay, pname, pscore, pnp, pown, pkx, pky, pkind, pks = flat(prediction.objects)
        gax, gay, gname, _, gnp, gown, gkx, gky, gkind, _ = flat(annotation.objects)
        DA = _dist_rows(pax, pay, gax, gay)

        def by_key(keys):
            out = {}
            for i, k in enumerate(keys):
                out.setdefault(k, []).append(i)
            return out

        def ranked(idx, score):                                         # score-descending, stable (np.argsort(-s, kind="stable"))
            return sorted(idx, key=lambda i: -score[i]) if len(idx) > 1 else idx

        def matched(rows, pred_groups, gt_groups, labels, score, inclusive, into):
            for label in labels:
                pl, gl = pred_groups.get(label, ()), gt_groups.get(label, ())
                tp, dists = _greedy_rows(rows, ranked(pl, score), gl, thr, inclusive)
                e = into[label]
                e.tp += tp; e.npos += len(gl); e.ndet += len(pl)
                if dists:
                    e.acc.extend(d / side for d in dists)

        p_by_name, g_by_name = by_key(pname), by_key(gname)
        matched(DA, p_by_name, g_by_name, self.labels, pscore, False, self.anchor_eval)            # evaluator.py:244-286
        if part_heatmap is not None:                                                                # evaluator.py:288-334
            rx, ry = [kp.x * fx for kp in part_heatmap], [kp.y * fy for kp in part_heatmap]
            matched(_dist_rows(rx, ry, gkx, gky), by_key([kp.kind for kp in part_heatmap]), by_key(gkind), self.kp_labels,
                    [kp.score for kp in part_heatmap], False, self.part_eval)
        if eval_classif:                                                                            # evaluator.py:427-474 (<= threshold)
            matched(DA, by_key([f"{n}_{k}" for n, k in zip(pname, pnp)]), by_key([f"{n}_{k}" for n, k in zip(gname, gnp)]),
                    Evaluator.get_classification_labels(), pscore, True, self.classification_eval)
        if eval_csi:                                                                                # evaluator.py:380-419, 538-581
            DP = _dist_rows(pkx, pky, gkx, gky)
            p_parts, g_parts = [{} for _ in pname], [{} for _ in gname]
            for k, i in enumerate(pown):
                p_parts[i].setdefault(pkind[k], []).append(k)
            for k, j in enumerate(gown):
                g_parts[j].setdefault(gkind[k], []).append(k)
            p_parts = [{kind: ranked(idx, pks) for kind, idx in d.items()} for d in p_parts]
            csi_thr = a.csi_threshold
            for label in self.labels:
                pl, gl = p_by_name.get(label, ()), g_by_name.get(label, ())
                res = self.csi_eval[label]
                res.ndet += len(pl); res.npos += len(gl)
                visited = set()
                for i in ranked(pl, pscore):
                    best, idx = 0.0, None
                    mine = p_parts[i]
                    for j in gl:
                        theirs = g_parts[j]
                        tp, npos, ndet = int(DA[i][j] < thr), 1, 1
                        for kind in theirs.keys() | mine.keys():
                            pk, gk = mine.get(kind, ()), theirs.get(kind, ())
                            npos += len(gk); ndet += len(pk)
                            tp += _greedy_rows(DP, pk, gk, thr, False)[0]
                        den = npos + ndet - tp
                        c = tp / den if den != 0 else 1
                        if c > best:
                            best, idx = c, j
                    if idx is not None and best >= csi_thr and idx not in visited:   # no match at all (every csi 0) is never a hit
                        visited.add(idx)
                        res.tp += 1
                        res.acc.append(best)

    def _accumulate_by_metric(self, prediction, annotation, part_heatmap=None, eval_csi=False, eval_classif=False):
        # both sides are mapped to image pixels ONCE per call (the four metrics only read them; each used to take its own resized copies:
        # seven deep copies per image)
        img_size = annotation.img_size
        conv = (self._to_image(annotation, img_size), self._to_image(prediction, img_size))
        self.anchor_eval += self.eval_anchor(prediction, annotation, _conv=conv)
        if part_heatmap is not None:
            self.part_eval += self.eval_part(annotation, part_heatmap, _conv=conv)
        if eval_csi:
            self.csi_eval += self.eval_csi(prediction, annotation, _conv=conv)
        if eval_classif:
            self.classification_eval += self.eval_classif(prediction, annotation, _conv=conv)

    def _to_image(self, ann, img_size):
        return ann.resized((self.args.width, self.args.height), img_size)

    def _match_objects(self, prediction, annotation, labels, key, inclusive, _conv=None):
        img_size = annotation.img_size
        annotation, prediction = _conv or (self._to_image(annotation, img_size), self._to_image(prediction, img_size))
        thr = min(img_size) * self.args.dist_threshold
        preds, gts = _by(prediction.objects, key), _by(annotation.objects, key)
        result = Evaluations(labels)
        for label in labels:
            pl, gl = preds.get(label, []), gts.get(label, [])
            tp, dists = greedy_nearest([(o.x, o.y) for o in pl], [o.anchor.score for o in pl], [(o.x, o.y) for o in gl], thr, inclusive)
            result[label] = Evaluation(tp, len(gl), len(pl), [d / min(img_size) for d in dists])
        return result

    def eval_anchor(self, prediction, annotation, _conv=None):                         # evaluator.py:244-286
        return self._match_objects(prediction, annotation, self.labels, lambda o: o.name, False, _conv)

    def eval_classif(self, prediction, annotation, _conv=None):                        # evaluator.py:427-474 (<= threshold)
        return self._match_objects(prediction, annotation, Evaluator.get_classification_labels(),
                                   lambda o: f"{o.name}_{o.nb_parts}", True, _conv)

    def eval_part(self, annotation, part_heatmap, _conv=None):                         # evaluator.py:288-334
        img_size = annotation.img_size
        annotation = _conv[0] if _conv else self._to_image(annotation, img_size)
        kps = [kp.resized((self.args.width, self.args.height), img_size) for kp in part_heatmap]
        thr = min(img_size) * self.args.dist_threshold
        preds = _by(kps, lambda kp: kp.kind)
        gts = _by((kp for obj in annotation.objects for kp in obj.parts), lambda kp: kp.kind)
        result = Evaluations(self.kp_labels)
        for label in self.kp_labels:
            pl, gl = preds.get(label, []), gts.get(label, [])
            tp, dists = greedy_nearest([(k.x, k.y) for k in pl], [k.score for k in pl], [(k.x, k.y) for k in gl], thr)
            result[label] = Evaluation(tp, len(gl), len(pl), [d / min(img_size) for d in dists])
        return result

    @staticmethod
    def compute_csi(prediction, target, dist_thresh):                                  # evaluator.py:538-581
        if prediction.name != target.name:
            return 0.0
        e = Evaluation(0, 1, 1)
        e.tp += int(prediction.distance(target) < dist_thresh)
        preds_kp, gts_kp = _by(prediction.parts, lambda kp: kp.kind), _by(target.parts, lambda kp: kp.kind)
        for kind in gts_kp.keys() | preds_kp.keys():
            pl, gl = preds_kp.get(kind, []), gts_kp.get(kind, [])
            e.npos += len(gl); e.ndet += len(pl)
            tp, _ = greedy_nearest([(k.x, k.y) for k in pl], [k.score for k in pl], [(k.x, k.y) for k in gl], dist_thresh)
            e.tp += tp
        return e.csi

    def eval_csi(self, prediction, annotation, _conv=None):                            # evaluator.py:380-419
        img_size = annotation.img_size
        annotation, prediction = _conv or (self._to_image(annotation, img_size), self._to_image(prediction, img_size))
        thr = min(img_size) * self.args.dist_threshold
        preds, gts = _by(prediction.objects, lambda o: o.name), _by(annotation.objects, lambda o: o.name)
        result = Evaluations(self.labels)
        for label in self.labels:
            pl, gl = preds.get(label, []), gts.get(label, [])
            res = result[label]
            res.ndet, res.npos = len(pl), len(gl)
            visited = np.zeros(len(gl), bool)
            for pred in sorted(pl, key=lambda o: o.anchor.score, reverse=True):
                best, idx = 0.0, None
                for j, gt in enumerate(gl):
                    c = Evaluator.compute_csi(pred, gt, thr)
                    if c > best:
                        best, idx = c, j
                if idx is not None and best >= self.args.csi_threshold and not visited[idx]:
                    visited[idx] = True
                    res.tp += 1
                    res.acc.append(best)
        return result

    # ------------------------------------------------------------------ reporting (evaluator.py:583-646)
    def _sections(self):
        return {"Anchor Location": self.anchor_eval, "Part Location": self.part_eval, "All Kps Location": self.kps_eval,
                "CSI": self.csi_eval, "Classification": self.classification_eval}

    def pretty_print(self):
        try:
            from rich import print as rprint
            from rich.table import Column, Table
        except ImportError:                                    # plain text when rich is absent
            print(repr(self))
            return
        for title, evals in self._sections().items():
            cols = [Column(h, justify="right", style="green" if h == "F1 Score" else None) for h in Evaluation.HEADERS]
            table = Table(Column("Label", style="bold"), *cols, title=title)
            for label, e in evals.items():
                table.add_row(label, *e.stats())
            if len(evals) > 1:
                table.add_row("Total", *evals.reduce().stats(), style="bold")
            rprint(table)

    def _csv_kps_str(self) -> str:
        evals = self.kps_eval
        return "\n".join(",".join((label, str(evals[label].recall), str(evals[label].precision), str(evals[label].f1_score),
                                   str(evals[label].avg_acc))) for label in sorted(evals.labels))

    def save_kps_csv(self, path: Path):
        Path(path).write_text(self._csv_kps_str())

    def __repr__(self):
        out = ""
        for name, evals in self._sections().items():
            out += f"{name}\n"
            if len(evals) > 1:
                out += f"  total: {evals.reduce()}\n"
            for label, e in sorted(evals.items(), key=lambda t: t[0]):
                out += f"  {label}: {e}\n"
        return out
