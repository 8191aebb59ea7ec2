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


# ---------------------------------------------------------------------------------------------
# BASELINE configs[0]: 16 synthetic samples on disk (PNG + JSON, README.md:40-71 schema with "box": null)
# ---------------------------------------------------------------------------------------------
EVAL16_LABELS = {"bean": 0, "maize": 1}
EVAL16_PARTS = {"leaf": 0}


def write_evaluate16_dir(g, directory):
    """Materialise the scenes of tests/golden/evaluate16.npz as img_NN.png + img_NN.json under `directory`
    (image sizes as recorded; blocky seeded-noise content) and return the 16 planted head tensors (7, 128, 128),
    carried by the golden in quantised form."""
    import json
    from pathlib import Path

    from PIL import Image

    directory = Path(directory)
    directory.mkdir(parents=True, exist_ok=True)
    W, H, M, N, K, P = (int(v) for v in g["cfg"])
    rl = {v: k for k, v in EVAL16_LABELS.items()}
    rp = {v: k for k, v in EVAL16_PARTS.items()}
    heads = []
    for n in range(16):
        iw, ih = (int(v) for v in g[f"size{n}"])
        objs = scene_from_flat(g[f"scene{n}_objs"], g[f"scene{n}_parts"])
        rng = np.random.default_rng(100 + n)
        small = rng.integers(0, 256, (ih // 16, iw // 16, 3), dtype=np.uint8)
        Image.fromarray(np.repeat(np.repeat(small, 16, 0), 16, 1)).save(directory / f"img_{n:02d}.png")
        js = {"image_path": str(directory / f"img_{n:02d}.png"), "img_size": [iw, ih],
              "objects": [{"label": rl[l], "box": None,
                           "parts": [{"kind": "stem", "location": {"x": x, "y": y}}]
                           + [{"kind": rp[k], "location": {"x": px, "y": py}} for (k, px, py) in ps]}
                          for (l, x, y, ps) in objs]}
        (directory / f"img_{n:02d}.json").write_text(json.dumps(js))
        # the planted head tensors travel in the fixture (exp / log differ by an ulp between host CPUs): int16 logits on a
        # 1/1024 grid, int8 regression channels on a 1/16 grid, the planted offset / embedding cells in full fp32
        reg = (g[f"head{n}_reg_q16"].astype(np.float32) / np.float32(16)).reshape(4, -1)
        reg[:, g[f"head{n}_cells"]] = g[f"head{n}_cell_vals"].T
        head = np.concatenate([g[f"head{n}_hm_q1024"].astype(np.float32) / np.float32(1024), reg.reshape(4, H // 4, W // 4)], 0)
        heads.append(head)
    return heads


def assert_evaluator_equals_golden(ev, g):
    for sec, evals in (("anchor", ev.anchor_eval), ("part", ev.part_eval), ("csi", ev.csi_eval), ("classif", ev.classification_eval)):
        assert list(evals.labels) == list(g[f"{sec}_labels"])
        counts = np.array([[e.tp, e.npos, e.ndet] for _, e in evals.items()], np.int64)
        np.testing.assert_array_equal(counts, g[f"{sec}_counts"], err_msg=sec)
        for label, e in evals.items():
            np.testing.assert_array_equal(np.array(e.acc, np.float64), g[f"{sec}_acc_{label}"], err_msg=f"{sec} {label}")
    assert ev._csv_kps_str() == str(g["csv"])


# ---------------------------------------------------------------------------------------------
# decoder parity: HIP packed result vs oracle.decode_tensors
# ---------------------------------------------------------------------------------------------
def safe_ranks(es, rel=2e-6):
    """Ranks whose score is separated from both neighbours by more than `rel` (relative): GPU and CPU sigmoids differ by a
    few ulp, so only there is the rank -> peak mapping defined identically on both sides."""
    es = np.asarray(es)
    gap = np.abs(np.diff(es.astype(np.float64), axis=1)) / np.maximum(es[:, 1:], 1e-30)
    safe = np.ones_like(es, bool)
    safe[:, 1:] &= gap > rel
    safe[:, :-1] &= gap > rel
    return safe


def assert_decode_matches_oracle(got, t, conf, sig_tol):
    """got = Decoder.split_packed(...) (numpy), t = oracle.decode_tensors(...).  Bit-exact where the ranking is defined:
      * per safe rank: flat index, class, refined x / y (same fp32 adds), gathered embedding; scores to `sig_tol`;
      * grouping, UNCONDITIONALLY for every safe part rank: the part is attached to the same anchor PEAK (class, flat index)
        or to none on both sides -- comparing the anchor's identity instead of its rank keeps the assertion meaningful when
        two near-tied anchors swap ranks;
      * in images whose above-threshold anchors are all safe, additionally the raw `assign` rank array.
    Returns (parts checked, parts total, images with the strict rank check)."""
    conf32 = np.float32(conf)
    safe = {}
    for grp in ("anchor", "part"):
        es = t[f"{grp}_out"][..., 2]
        s = safe[grp] = safe_ranks(es)
        np.testing.assert_array_equal(got[f"{grp}_ind"][s], t[f"{grp}_inds"][s])
        for ch in (3, 0, 1):
            np.testing.assert_array_equal(got[f"{grp}_out"][..., ch][s], t[f"{grp}_out"][..., ch][s])
        np.testing.assert_allclose(got[f"{grp}_out"][..., 2], es, **sig_tol)
    ps = safe["part"]
    np.testing.assert_array_equal(got["part_emb"][ps], t["part_embeddings"][ps])
    np.testing.assert_array_equal(got["part_out"][..., 4:6][ps], t["part_out"][..., 4:6][ps])
    B, P = ps.shape
    want_assign = np.where(t["valid"], t["min_inds"], -1)
    bi = np.arange(B)[:, None]

    def identity(assign, a_ind, a_cls):
        a = np.maximum(assign, 0)
        ident = a_cls[bi, a].astype(np.int64) * (1 << 32) + a_ind[bi, a].astype(np.int64)
        return np.where(assign >= 0, ident, -1)

    id_got = identity(got["assign"], got["anchor_ind"], got["anchor_out"][..., 3])
    id_want = identity(want_assign, t["anchor_inds"], t["anchor_out"][..., 3])
    np.testing.assert_array_equal(id_got[ps], id_want[ps])
    strict = 0
    for b in range(B):
        live = t["anchor_out"][b, :, 2] > conf32
        if safe["anchor"][b][live].all():
            np.testing.assert_array_equal(got["assign"][b][ps[b]], want_assign[b][ps[b]])
            strict += 1
    return int(ps.sum()), int(ps.size), strict


# ---------------------------------------------------------------------------------------------
# per-layer operands of an oracle run (tests/test_gpu_production_parity.py)
# ---------------------------------------------------------------------------------------------
def oracle_conv_trace(ref, x, dhead_of, device, skip=()):
    """Run `ref(x)` (torch-CPU oracle network, training mode) and its backward from `dhead_of(head)`, capturing for every
    Conv2d the tensors that layer saw: input x, output y, output gradient dy and input gradient dx (None where the input needs
    no gradient), each moved to `device` as a contiguous NHWC tensor.  Returns (head.detach(), {module name: dict})."""
    import torch
    trace, handles = {}, []

    def nhwc(t):
        return t.detach().permute(0, 2, 3, 1).contiguous().to(device)

    for name, m in ref.named_modules():
        if not isinstance(m, torch.nn.Conv2d) or name in skip:
            continue
        rec = trace[name] = {"module": m}

        def fwd_hook(mod, inp, out, rec=rec):
            rec["x"], rec["y"] = nhwc(inp[0]), nhwc(out)

        def bwd_hook(mod, gin, gout, rec=rec):
            rec["dy"] = nhwc(gout[0])
            rec["dx"] = nhwc(gin[0]) if gin[0] is not None else None

        handles.append(m.register_forward_hook(fwd_hook))
        handles.append(m.register_full_backward_hook(bwd_hook))
    try:
        head = ref(x)
        head.backward(dhead_of(head))
    finally:
        for h in handles:
            h.remove()
    return head.detach(), trace
