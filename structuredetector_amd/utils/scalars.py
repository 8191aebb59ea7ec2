"""Training scalars (src/sdnet/model/trainer.py:34,126-133,240-256: a TensorBoard `SummaryWriter`).

TensorBoard is an optional import: when `torch.utils.tensorboard` loads, scalars go to an event file exactly as the reference's do
(same tags, same `global_step` numbering); they are ALSO always appended to `<log_dir>/scalars.jsonl`, one JSON object per
`add_scalar(s)` call, so a run leaves a readable record on hosts without TensorBoard (this image has none).  The image / heatmap
panels the reference draws at validation time (trainer.py:258-309) are not written.
"""
from __future__ import annotations

import json
from pathlib import Path


class ScalarWriter:
    def __init__(self, log_dir=None, tensorboard=True):
        self.log_dir = Path(log_dir) if log_dir is not None else None
        self._file = None
        self._tb = None
        self.backend = "none"
        if self.log_dir is None:
            return
        self.log_dir.mkdir(parents=True, exist_ok=True)
        self._file = open(self.log_dir / "scalars.jsonl", "a", buffering=1)
        self.backend = "jsonl"
        if tensorboard:
            try:
                from torch.utils.tensorboard import SummaryWriter
                self._tb = SummaryWriter(log_dir=str(self.log_dir))
                self.backend = "jsonl+tensorboard"
            except Exception:                                      # tensorboard not installed: the JSONL record is the log
                self._tb = None

    def add_scalar(self, tag, value, step):
        if self._file is not None:
            self._file.write(json.dumps({"tag": tag, "step": int(step), "value": float(value)}) + "\n")
        if self._tb is not None:
            self._tb.add_scalar(tag, float(value), int(step))

    def add_scalars(self, tag, values, step):
        values = {str(k): float(v) for k, v in values.items()}
        if self._file is not None:
            self._file.write(json.dumps({"tag": tag, "step": int(step), "values": values}) + "\n")
        if self._tb is not None:
            self._tb.add_scalars(tag, values, int(step))

    def flush(self):
        if self._file is not None:
            self._file.flush()
        if self._tb is not None:
            self._tb.flush()

    def close(self):
        self.flush()
        if self._file is not None:
            self._file.close(); self._file = None
        if self._tb is not None:
            self._tb.close(); self._tb = None


def metric_dicts(evaluator):
    """The eleven `add_scalars` dicts of the reference's validation pass (trainer.py:173-224, tags :243-256): per-label value + "total"."""
    def table(evals, attr):
        out = {label: getattr(e, attr) for label, e in evals.items()}
        out["total"] = getattr(evals.reduce(), attr)
        return out
    kps = evaluator.kps_eval
    return {
        "Metrics_AllKps/Precison": table(kps, "precision"),          # sic (trainer.py:244)
        "Metrics_AllKps/Recall": table(kps, "recall"),
        "Metrics_AllKps/F1": table(kps, "f1_score"),
        "Metrics_Anchor/Precision": table(evaluator.anchor_eval, "precision"),
        "Metrics_Anchor/Recall": table(evaluator.anchor_eval, "recall"),
        "Metrics_Anchor/f1": table(evaluator.anchor_eval, "f1_score"),
        "Metrics_Parts/Precision": table(evaluator.part_eval, "precision"),
        "Metrics_Parts/Recall": table(evaluator.part_eval, "recall"),
        "Metrics_Parts/f1": table(evaluator.part_eval, "f1_score"),
        "Metrics_CSI/f1": table(evaluator.csi_eval, "f1_score"),
        "Metrics_Classif/f1": table(evaluator.classification_eval, "f1_score"),
    }
