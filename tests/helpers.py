"""Shared helpers for the parity tests (data plumbing only)."""
import numpy as np


def scene_from_flat(objs, parts):
    """Inverse of gen_goldens.flat_scene: -> [(label, x, y, [(kind, x, y), ...]), ...]."""
    out, j = [], 0
    for (l, x, y, n) in objs:
        ps = [(int(parts[j + i][0]), float(parts[j + i][1]), float(parts[j + i][2])) for i in range(int(n))]
        j += int(n)
        out.append((int(l), float(x), float(y), ps))
    return out


def objects_to_arrays(objs):
    """oracle.assemble_objects output -> (n,4) [label,x,y,score], (m,5) [obj,kind,x,y,score]."""
    o = np.array([[l, a[0], a[1], a[2]] for (l, a, _) in objs], np.float64).reshape(-1, 4)
    p = np.array([[i, k, x, y, s] for i, (_, _, ps) in enumerate(objs) for (k, x, y, s) in ps], np.float64).reshape(-1, 5)
    return o, p


ENC_KEYS = ["anchor_hm", "part_hm", "anchor_inds", "part_inds", "anchor_offsets", "part_offsets",
            "embeddings", "anchor_mask", "part_mask"]
