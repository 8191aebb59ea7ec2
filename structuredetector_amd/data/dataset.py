"""Batch assembly helpers.  `collate_fn` keeps the reference's contract
(src/sdnet/data/dataset.py:58-87): stack every Encode field along a new leading dim, keep
`annotation` as a list."""
import torch

_TENSOR_KEYS = ["image", "anchor_hm", "part_hm", "anchor_offsets", "part_offsets", "embeddings", "anchor_inds",
                "part_inds", "anchor_mask", "part_mask"]


def collate_fn(elements):
    batch = {k: torch.stack([e[k] for e in elements], dim=0) for k in _TENSOR_KEYS}
    batch["annotation"] = [e["annotation"] for e in elements]
    return batch
