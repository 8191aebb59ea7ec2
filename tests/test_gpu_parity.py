"""GPU parity tests: the HIP path (through the C ABI) against the reference's golden vectors and
against the CPU oracle on seeded inputs.  Integer / index results are compared bit-exact;
floating-point maps to the tolerance stated at each assert (north_star: 1e-4 fp32)."""
import numpy as np
import pytest
import torch

from oracle import sdnet_oracle as O
from tests.helpers import ENC_KEYS, assert_decode_matches_oracle, objects_to_arrays, safe_ranks, scene_from_flat
from tests.test_host_cpu import make_args, to_annotation

pytestmark = pytest.mark.gpu
DEV = "cuda"
SIG_TOL = dict(rtol=4e-7, atol=0)      # clamped sigmoid: <= ~3 ulp between GPU expf/div and the CPU's


def dev(x):
    return torch.from_numpy(np.ascontiguousarray(x)).to(DEV)


def head_views(head, M, N):
    return {"anchor_hm": head[:, :M], "part_hm": head[:, M:M + N], "offsets": head[:, M + N:M + N + 2],
            "embeddings": head[:, M + N + 2:M + N + 4]}


def annotation_arrays(args, ann):
    objs = [(args.labels[o.name], (o.anchor.x, o.anchor.y, o.anchor.score),
             [(args.parts[k.kind], k.x, k.y, k.score) for k in o.parts]) for o in ann.objects]
    return objects_to_arrays(objs)


# ------------------------------------------------------------------------------------------ prims
def test_prims_vs_golden(golden_dir):
    from structuredetector_amd.utils import clamped_sigmoid, hypot, nms, topk, transpose_and_gather
    g = np.load(golden_dir / "prims.npz")
    np.testing.assert_allclose(clamped_sigmoid(dev(g["logits"])).cpu().numpy(), g["sig"], **SIG_TOL)
    np.testing.assert_array_equal(nms(dev(g["sig"])).cpu().numpy(), g["nms"])            # compare/select only: exact
    for k in (2, 7, 40):
        s, i, c, y, x = (t.cpu().numpy() for t in topk(dev(g["dense"]), k))
        np.testing.assert_array_equal(s, g[f"topk{k}_score"]); np.testing.assert_array_equal(i, g[f"topk{k}_ind"])
        np.testing.assert_array_equal(c, g[f"topk{k}_cls"]); np.testing.assert_array_equal(y, g[f"topk{k}_y"])
        np.testing.assert_array_equal(x, g[f"topk{k}_x"])
        assert i.dtype == np.int64
    np.testing.assert_array_equal(transpose_and_gather(dev(g["feat"]), dev(g["gind"])).cpu().numpy(), g["gathered"])
    from structuredetector_amd.utils import gather
    flat = dev(g["feat"][:, 0].reshape(2, -1))
    np.testing.assert_array_equal(gather(flat, dev(g["gind"])).cpu().numpy(), np.take_along_axis(g["feat"][:, 0].reshape(2, -1), g["gind"], 1))
    np.testing.assert_array_equal(hypot(dev(g["hyp_in"])).cpu().numpy(), g["hyp_out"])   # mul, add, sqrt: exact, no FMA


def test_topk_ties_and_zero_fill():
    """Order contract: score desc, class asc, flat index asc; zero slots = lowest non-peak indices."""
    from structuredetector_amd.utils import decode_peaks, topk
    x = np.zeros((1, 2, 8, 8), np.float32)
    x[0, 1, 3, 3] = 2.0; x[0, 0, 5, 5] = 2.0; x[0, 0, 1, 1] = 2.0; x[0, 1, 0, 0] = 3.0
    s, i, c, _, _ = (t.cpu().numpy() for t in topk(dev(x), 6))
    es, ei, ec, _, _ = O.topk(x, 6)
    np.testing.assert_array_equal(s, es); np.testing.assert_array_equal(i, ei); np.testing.assert_array_equal(c, ec)
    # fused path on logits with only three peaks: remaining slots are zeros at flat 0,1,2,... skipping peaks
    lg = np.full((2, 2, 16, 24), -20.0, np.float32)           # sigmoid clamps to 1e-6 everywhere: one big plateau
    s, i, c, y, xx = (t.cpu().numpy() for t in decode_peaks(dev(lg), 5))
    es, ei, ec, ey, ex = O.topk(O.nms(O.clamped_sigmoid(lg)), 5)
    np.testing.assert_array_equal(i, ei); np.testing.assert_array_equal(c, ec); np.testing.assert_allclose(s, es, **SIG_TOL)
    lg[:, :, :, :] = np.random.default_rng(0).standard_normal(lg.shape).astype(np.float32) * 0.01 - 30.0   # all clamp to 1e-6
    lg[0, 1, 4, 4] = 3.0; lg[0, 0, 0, 0] = 2.0; lg[1, 0, 15, 23] = 1.0
    s, i, c, y, xx = (t.cpu().numpy() for t in decode_peaks(dev(lg), 7))
    es, ei, ec, ey, ex = O.topk(O.nms(O.clamped_sigmoid(lg)), 7)
    np.testing.assert_array_equal(i, ei); np.testing.assert_array_equal(c, ec)
    np.testing.assert_array_equal(y, ey); np.testing.assert_array_equal(xx, ex)


@pytest.mark.parametrize("shape,k", [((3, 2, 40, 56), 17), ((2, 3, 128, 128), 40), ((1, 8, 256, 256), 512), ((2, 1, 33 * 4, 20), 1)])
def test_decode_peaks_vs_oracle_noise(shape, k):
    """White-noise logits: ~4% of the pixels survive the NMS, exercising the radix-select path."""
    from structuredetector_amd.utils import decode_peaks
    rng = np.random.default_rng(shape[2] * 7 + k)
    lg = rng.standard_normal(shape).astype(np.float32)
    s, i, c, y, x = (t.cpu().numpy() for t in decode_peaks(dev(lg), k))
    es, ei, ec, ey, ex = O.topk(O.nms(O.clamped_sigmoid(lg)), k)
    # GPU and CPU sigmoids may differ by a few ulp: indices must agree wherever the oracle's ranking has margin
    gap = np.abs(np.diff(es.astype(np.float64), axis=1)) / es[:, 1:]
    safe = np.ones_like(es, bool)
    safe[:, 1:] &= gap > 2e-6; safe[:, :-1] &= gap > 2e-6
    assert safe.mean() > 0.6
    np.testing.assert_array_equal(i[safe], ei[safe]); np.testing.assert_array_equal(c[safe], ec[safe])
    np.testing.assert_array_equal(y[safe], ey[safe]); np.testing.assert_array_equal(x[safe], ex[safe])
    np.testing.assert_allclose(s, es, **SIG_TOL)
    assert (np.diff(s, axis=1) <= 0).all()


# ------------------------------------------------------------------------------------------ decoder
@pytest.mark.parametrize("tag", ["scene_cfg512", "scene_small256"])
def test_decoder_vs_golden(golden_dir, tag):
    from structuredetector_amd.data import Decoder
    g = np.load(golden_dir / f"{tag}.npz")
    W, H, M, N, K, P = (int(v) for v in g["cfg"])
    args = make_args(M, N, K, P)
    head = dev(g["head"])
    md = Decoder(args)(head_views(head, M, N), return_metadata=True)
    for grp, key in (("anchor", "topk_anchor"), ("part", "topk_kp")):
        s, i, c, y, x = (t.cpu().numpy() for t in md[key])
        ref_s = g[f"dec_{grp}_score"]
        pos = ref_s > 0                       # above-threshold peaks (masked ones are -1 in the reference too)
        assert i.dtype == np.int64
        np.testing.assert_array_equal(s > 0, pos)
        np.testing.assert_array_equal(i[pos], g[f"dec_{grp}_ind"][pos])          # bit-exact indices
        np.testing.assert_array_equal(c[pos], g[f"dec_{grp}_cls"][pos])
        np.testing.assert_array_equal(x[pos], g[f"dec_{grp}_x"][pos])            # refined coords: same fp32 add
        np.testing.assert_array_equal(y[pos], g[f"dec_{grp}_y"][pos])
        np.testing.assert_allclose(s[pos], ref_s[pos], **SIG_TOL)
    pos = g["dec_part_score"] > 0
    np.testing.assert_array_equal(md["embeddings"].cpu().numpy()[pos], g["dec_embeddings"][pos])
    for b in range(head.shape[0]):
        o, p = annotation_arrays(args, md["annotation"][b])
        ro, rp = g[f"ann{b}_objs"], g[f"ann{b}_parts"]
        assert o.shape == ro.shape and p.shape == rp.shape
        np.testing.assert_array_equal(o[:, :3], ro[:, :3]); np.testing.assert_allclose(o[:, 3], ro[:, 3], **SIG_TOL)
        np.testing.assert_array_equal(p[:, :4], rp[:, :4]); np.testing.assert_allclose(p[:, 4], rp[:, 4], **SIG_TOL)   # grouping: exact
        r = np.array([[args.parts[k.kind], k.x, k.y, k.score] for k in md["raw_parts"][b]], np.float64).reshape(-1, 4)
        rr = g[f"raw{b}"]
        assert r.shape == rr.shape
        np.testing.assert_array_equal(r[:, :3], rr[:, :3])
    assert md["annotation"][0].objects and md["annotation"][0].image_name == "batch_0"
    # plain call (fast selection: peaks <= conf dropped before the sort) returns exactly the same annotations
    anns = Decoder(args)(head_views(head, M, N))
    assert [len(a) for a in anns] == [len(a) for a in md["annotation"]]
    for a, bref in zip(anns, md["annotation"]):
        ao, ap = annotation_arrays(args, a)
        bo, bp = annotation_arrays(args, bref)
        np.testing.assert_array_equal(ao, bo); np.testing.assert_array_equal(ap, bp)


@pytest.mark.parametrize("B,img,M,N,K,P", [(5, 256, 2, 1, 20, 40), (2, 512, 8, 8, 128, 512), (3, 128, 1, 1, 3, 2)])
def test_decoder_vs_oracle_random_scenes(B, img, M, N, K, P):
    from structuredetector_amd.data import Decoder
    rng = np.random.default_rng(B * 1000 + img)
    heads = []
    for _ in range(B):
        objs = O.synthetic_scene(rng, img, img, M, N, 4, 30 if K > 20 else 10)
        e = O.encode(img, img, objs, M, N, K, P, 4.0, 0.1)
        heads.append(O.head_from_targets(rng, e, M, N, noise=0.3))
    head = np.stack(heads)
    args = make_args(M, N, K, P)
    dec = Decoder(args)
    packed, (b_, k_, p_, h, w) = dec.decode_packed(head_views(dev(head), M, N), 0.5, 0.1)
    got = dec.split_packed(packed.cpu().numpy(), B, K, P)
    t = O.decode_tensors(head[:, :M], head[:, M:M + N], head[:, M + N:M + N + 2], head[:, M + N + 2:], K, P, 0.5, 0.1)
    checked, total, strict = assert_decode_matches_oracle(got, t, 0.5, SIG_TOL)
    assert checked >= 0.9 * total and (t["valid"].sum() > 0 or K < 4)


def test_decoder_dense_stress_scenes_grouping_vs_oracle():
    """BASELINE configs[4] workload: 1024x1024, 8 labels / 8 parts, K=128, P=512, 64-96 objects per image (dense scenes:
    >= 11 k NMS survivors per list -> the radix-select path), targets rendered by the product Encode, head synthesised from
    them, sd_decode vs the oracle: indices on the safe ranks and the GROUPING of every safe part (no all-or-nothing guard)."""
    from structuredetector_amd.data import Decoder, Encode
    from structuredetector_amd.data.synthetic import synthetic_batch
    B, img, M, N, K, P = 4, 1024, 8, 8, 128, 512
    args = make_args(M, N, K, P, device=torch.device(DEV))
    rng = np.random.default_rng(4096)
    enc = Encode(args)
    tgt = enc.render(enc.plan(img, img, *synthetic_batch(rng, B, img, img, M, N, 64, 96)), DEV)
    tgt = {k: v.cpu().numpy() for k, v in tgt.items() if isinstance(v, torch.Tensor)}
    assert tgt["anchor_mask"].sum(1).min() >= 64
    head = np.stack([O.head_from_targets(rng, {k: v[b] for k, v in tgt.items()}, M, N, noise=0.3) for b in range(B)])
    dec = Decoder(args)
    t = O.decode_tensors(head[:, :M], head[:, M:M + N], head[:, M + N:M + N + 2], head[:, M + N + 2:], K, P, 0.5, 0.1)
    packed, _ = dec.decode_packed(head_views(dev(head), M, N), 0.5, 0.1, exact_topk=True)
    got = dec.split_packed(packed.cpu().numpy(), B, K, P)
    checked, total, strict = assert_decode_matches_oracle(got, t, 0.5, SIG_TOL)
    attached = int(t["valid"].sum())
    assert checked >= 0.95 * total, (checked, total)
    assert attached >= 64 * B, attached                           # really dense: hundreds of part -> anchor links asserted
    # Decoder.__call__ without metadata (fast selection: peaks below fp32(conf) never enter the sort) -> the same objects with
    # the same parts.  Compared as {anchor (label, x, y) -> sorted parts (kind, x, y)}: independent of the order two near-tied
    # peaks take in the lists, so it is asserted for EVERY image; coordinates are the same fp32 adds on both sides -> exact.
    anns = dec(head_views(dev(head), M, N))
    n_links = 0
    for b in range(B):
        o, p = annotation_arrays(args, anns[b])
        ro, rp = objects_to_arrays(O.assemble_objects(t, b, 0.5, 4.0, img // 4, img // 4))
        assert o.shape == ro.shape and p.shape == rp.shape and len(o) >= 48, (o.shape, ro.shape, p.shape, rp.shape)

        def table(objs, parts):
            return {tuple(objs[i, :3]): sorted(tuple(q[1:4]) for q in parts[parts[:, 0] == i]) for i in range(len(objs))}

        got_t, want_t = table(o, p), table(ro, rp)
        assert len(got_t) == len(o) and got_t == want_t, f"image {b}: grouping differs"
        n_links += len(p)
    assert n_links >= 64 * B


def test_decode_group_matches_fused():
    """sd_decode_group fed with sd_decode_peaks outputs == the fused sd_decode."""
    import ctypes
    from structuredetector_amd import _lib as L
    from structuredetector_amd.data import Decoder
    from structuredetector_amd.utils import decode_peaks
    rng = np.random.default_rng(5)
    B, M, N, K, P, h, w = 3, 2, 2, 9, 14, 48, 64
    head = dev((2 * rng.standard_normal((B, M + N + 4, h, w))).astype(np.float32))
    v = head_views(head, M, N)
    dec = Decoder(make_args(M, N, K, P))
    packed, _ = dec.decode_packed(v, 0.5, 0.1)
    a = decode_peaks(v["anchor_hm"], K); p = decode_peaks(v["part_hm"], P)
    packed2 = torch.empty_like(packed)
    o, o_p, o_sb, o_sc = L.map_view(v["offsets"]); e, e_p, e_sb, e_sc = L.map_view(v["embeddings"])
    L.check(L.lib().sd_decode_group(a[0].data_ptr(), a[1].data_ptr(), a[2].data_ptr(), p[0].data_ptr(), p[1].data_ptr(),
                                    p[2].data_ptr(), o_p, o_sb, o_sc, e_p, e_sb, e_sc, B, h, w, K, P,
                                    float(np.float32(0.5)), float(np.float32(0.1 * min(h, w))), packed2.data_ptr(), L.stream()))
    assert torch.equal(packed, packed2)


# ------------------------------------------------------------------------------------------ encode
@pytest.mark.parametrize("tag", ["scene_cfg512", "scene_small256"])
def test_encode_vs_golden(golden_dir, tag):
    from structuredetector_amd.data import Encode
    g = np.load(golden_dir / f"{tag}.npz")
    W, H, M, N, K, P = (int(v) for v in g["cfg"])
    args = make_args(M, N, K, P, device=torch.device(DEV))
    n_img = g["head"].shape[0]
    anns = [to_annotation(args, scene_from_flat(g[f"scene{n}_objs"], g[f"scene{n}_parts"])) for n in range(n_img)]
    out = Encode(args).batch((W, H), anns)
    for k in ENC_KEYS:
        ref = np.stack([g[f"enc{n}_{k}"] for n in range(n_img)])
        got = out[k].cpu().numpy()
        assert got.dtype == ref.dtype and got.shape == ref.shape, k
        if k.endswith("_hm"):
            np.testing.assert_allclose(got, ref, rtol=1e-5, atol=1e-7, err_msg=k)     # expf: few ulp (budget 1e-4)
            np.testing.assert_array_equal(got == 1.0, ref == 1.0)                    # peaks are exactly 1.0 (focal loss needs it)
        else:
            np.testing.assert_array_equal(got, ref, err_msg=k)
    # single-sample form: same keys / shapes as the reference's Encode.__call__
    one = Encode(args)(torch.zeros(3, H, W), anns[0])
    assert set(one) == set(ENC_KEYS) | {"image", "annotation"}
    assert one["anchor_hm"].shape == (M, H // 4, W // 4) and one["part_inds"].shape == (P,)
    assert one["anchor_mask"].dtype == torch.bool and one["anchor_inds"].dtype == torch.int64


def test_encode_truncation_vs_golden(golden_dir):
    from structuredetector_amd.data import Encode
    g = np.load(golden_dir / "encode_trunc.npz")
    W, H, M, N, K, P = (int(v) for v in g["cfg"])
    args = make_args(M, N, K, P, device=torch.device(DEV))
    for name in g["cases"]:
        ann = to_annotation(args, scene_from_flat(g[f"{name}_objs"], g[f"{name}_parts"]))
        out = Encode(args)(torch.zeros(3, H, W), ann)
        for k in ENC_KEYS:
            if k.endswith("_hm"):
                np.testing.assert_allclose(out[k].cpu().numpy(), g[f"{name}_{k}"], rtol=1e-5, atol=1e-7, err_msg=f"{name} {k}")
            else:
                np.testing.assert_array_equal(out[k].cpu().numpy(), g[f"{name}_{k}"], err_msg=f"{name} {k}")


# ------------------------------------------------------------------------------------------ loss
@pytest.mark.parametrize("tag", ["scene_cfg512", "scene_small256"])
@pytest.mark.parametrize("fn", ["mse", "focal"])
@pytest.mark.parametrize("as_views", [True, False])
def test_loss_vs_golden(golden_dir, tag, fn, as_views):
    from structuredetector_amd.model import Loss
    g = np.load(golden_dir / f"{tag}.npz")
    W, H, M, N, K, P = (int(v) for v in g["cfg"])
    n_img = g["head"].shape[0]
    target = {k: dev(np.stack([g[f"enc{n}_{k}"] for n in range(n_img)])) for k in ENC_KEYS}
    head = dev(g["head"]).requires_grad_(True)
    if as_views:
        inp = head_views(head * 1.0, M, N)                 # non-leaf base -> slices are views of one tensor
    else:
        inp = {k: v.clone() for k, v in head_views(head, M, N).items()}
    crit = Loss(make_args(M, N, K, P, hm_loss_fn=fn))
    val = crit(inp, target)
    val.backward()
    ref = g[f"loss_{fn}"]
    got = [val.item(), float(crit.stats.hm_loss), float(crit.stats.offset_loss), float(crit.stats.embedding_loss)]
    np.testing.assert_allclose(got, ref, rtol=1e-4, atol=1e-7)                        # north_star: 1e-4
    assert abs(float(crit.stats.total_loss) - ref[0]) <= 1e-4 * abs(ref[0]) + 1e-7
    gref = g[f"lossgrad_{fn}"]
    np.testing.assert_allclose(head.grad.cpu().numpy(), gref, rtol=1e-4, atol=1e-4 * np.abs(gref).max())


def test_loss_rejects_host_targets_and_bad_shapes():
    from structuredetector_amd import _lib as L
    from structuredetector_amd.model import Loss
    e = O.collate([O.encode(64, 96, [], 2, 1, 4, 6, 4.0, 0.1)])
    head = torch.zeros(1, 7, 24, 16, device=DEV)
    crit = Loss(make_args(2, 1, 4, 6))
    with pytest.raises(L.SdError):
        crit(head_views(head, 2, 1), {k: torch.from_numpy(v) for k, v in e.items()})            # targets left on the host
    bad = {k: dev(v) for k, v in e.items()}
    bad["part_inds"] = bad["part_inds"][:, :3]
    with pytest.raises(L.SdError):
        crit(head_views(head, 2, 1), bad)


def test_loss_degenerate_batches():
    """No valid keypoints (L1 terms -> 0, loss.py:59-61) and no positives (focal -> -neg, loss.py:110)."""
    from structuredetector_amd.model import Loss
    rng = np.random.default_rng(3)
    B, M, N, K, P, h, w = 2, 2, 1, 4, 6, 16, 24
    e = O.collate([O.encode(w * 4, h * 4, [], M, N, K, P, 4.0, 0.1) for _ in range(B)])
    head = (rng.standard_normal((B, M + N + 4, h, w))).astype(np.float32)
    for fn in ("mse", "focal"):
        r = O.loss(head, e, M, N, hm_loss_fn=fn, want_grad=True)
        x = dev(head).requires_grad_(True)
        crit = Loss(make_args(M, N, K, P, hm_loss_fn=fn))
        val = crit(head_views(x * 1.0, M, N), {k: dev(v) for k, v in e.items()})
        val.backward()
        np.testing.assert_allclose(val.item(), r["total"], rtol=1e-4)
        assert float(crit.stats.offset_loss) == 0.0 and float(crit.stats.embedding_loss) == 0.0
        np.testing.assert_allclose(x.grad.cpu().numpy(), r["grad"], rtol=1e-4, atol=1e-4 * np.abs(r["grad"]).max())


# ------------------------------------------------------------------------------------------ full-size properties
def test_roundtrip_full_size_batch():
    """BASELINE config sizes (bs=64, 512x512, 2 labels / 1 part): Encode -> synthesised head -> Decoder recovers
    every object and part exactly (the reference's own round-trip identity)."""
    from structuredetector_amd.data import Decoder, Encode
    B, img, M, N, K, P = 64, 512, 2, 1, 20, 40
    args = make_args(M, N, K, P, device=torch.device(DEV))
    rng = np.random.default_rng(77)
    scenes = []
    for _ in range(B):
        # keep keypoints >= 6 output pixels apart so that every one is its own 5x5 maximum
        objs, taken = [], []
        for (l, x, y, ps) in O.synthetic_scene(rng, img, img, M, N, 6, 12, 1, 2):
            def free(px, py):
                return all(max(abs(px // 4 - qx), abs(py // 4 - qy)) >= 6 for qx, qy in taken)
            if not free(x, y):
                continue
            taken.append((x // 4, y // 4))
            kept = []
            for (k, px, py) in ps:
                if free(px, py):
                    taken.append((px // 4, py // 4)); kept.append((k, px, py))
            objs.append((l, x, y, kept))
        scenes.append(objs)
    anns = [to_annotation(args, s) for s in scenes]
    tgt = Encode(args).batch((img, img), anns)
    hm = torch.cat([tgt["anchor_hm"], tgt["part_hm"]], 1).clamp(1e-4, 1 - 1e-4)
    logits = torch.log(hm / (1 - hm))
    h = w = img // 4
    reg = torch.zeros(B, 4, h * w, device=DEV)
    bi = torch.arange(B, device=DEV)[:, None]
    am, pm = tgt["anchor_mask"], tgt["part_mask"]
    reg[bi.expand_as(am)[am], 0, tgt["anchor_inds"][am]] = tgt["anchor_offsets"][am][:, 0]
    reg[bi.expand_as(am)[am], 1, tgt["anchor_inds"][am]] = tgt["anchor_offsets"][am][:, 1]
    reg[bi.expand_as(pm)[pm], 0, tgt["part_inds"][pm]] = tgt["part_offsets"][pm][:, 0]
    reg[bi.expand_as(pm)[pm], 1, tgt["part_inds"][pm]] = tgt["part_offsets"][pm][:, 1]
    reg[bi.expand_as(pm)[pm], 2, tgt["part_inds"][pm]] = tgt["embeddings"][pm][:, 0]
    reg[bi.expand_as(pm)[pm], 3, tgt["part_inds"][pm]] = tgt["embeddings"][pm][:, 1]
    head = torch.cat([logits, reg.view(B, 4, h, w)], 1)
    anns_out = Decoder(args)(head_views(head, M, N))
    n_obj = n_part = 0

    def canon(ann):
        return sorted(((o.name, o.x, o.y, sorted((p.kind, p.x, p.y) for p in o.parts)) for o in ann.objects),
                      key=lambda t: (t[0], round(t[1], 1), round(t[2], 1)))

    for b in range(B):
        want, got = canon(anns[b]), canon(anns_out[b])
        assert len(want) == len(got), f"image {b}"
        for wo, go in zip(want, got):
            assert wo[0] == go[0] and abs(wo[1] - go[1]) < 1e-3 and abs(wo[2] - go[2]) < 1e-3, f"image {b}"
            assert len(wo[3]) == len(go[3]), f"image {b}: grouping differs"
            for wp, gp in zip(wo[3], go[3]):
                assert wp[0] == gp[0] and abs(wp[1] - gp[1]) < 1e-3 and abs(wp[2] - gp[2]) < 1e-3, f"image {b}"
        n_obj += len(want); n_part += sum(len(o[3]) for o in want)
    assert n_obj > 300 and n_part > 300


def test_nms_idempotent_and_sorted_full_size():
    from structuredetector_amd.utils import clamped_sigmoid, decode_peaks, nms
    x = torch.randn(64, 3, 128, 128, device=DEV, generator=torch.Generator(DEV).manual_seed(1))
    s = clamped_sigmoid(x)
    n1 = nms(s)
    assert torch.equal(nms(n1), n1)                                  # idempotence
    frac = (n1 > 0).float().mean().item()
    assert 0.03 < frac < 0.05                                        # ~4.1 % of white-noise pixels survive
    sc, ind, cls, ys, xs = decode_peaks(x, 40)
    assert (sc[:, 1:] <= sc[:, :-1]).all()                           # sortedness
    flat = n1.view(64, -1)
    assert torch.equal(sc, flat.topk(40, dim=1).values)              # same score multiset as a dense top-k of the NMS map
    picked = n1.view(64, 3, -1)[torch.arange(64, device=DEV)[:, None], cls.long(), ind]
    assert torch.equal(picked, sc)                                   # (cls, ind) really address those scores
    assert torch.equal(ys * 128 + xs, ind.float())


# ------------------------------------------------------------------------------------------ thresholds that fp32 cannot represent
def test_decoder_thresholds_vs_golden(golden_dir):
    """conf 0.4 (fp32 0.4000000060) / dist 0.1*64 (fp32 6.4000000954) with scores and distances planted exactly ON the
    rounded thresholds (tests/golden/decode_thresholds.npz, produced by the reference's Decoder): the anchor whose score
    equals fp32(conf) is masked for the association yet emitted as a part-less object, the part with that score stays in
    raw_parts, the part exactly fp32(dist) away is not attached and the one an ulp closer is -- in BOTH selection modes
    (`return_metadata=True`: exact top-k; plain call: peaks below fp32(conf) dropped before the sort, `>=` keeps equality)."""
    from structuredetector_amd.data import Decoder
    from structuredetector_amd.utils import clamped_sigmoid
    g = np.load(golden_dir / "decode_thresholds.npz")
    W, H, M, N, K, P = (int(v) for v in g["cfg"])
    conf, dist = float(g["conf"]), float(g["dist"])
    head = g["head"].copy()
    c32 = np.float32(conf)
    # the planted logit must give EXACTLY fp32(0.4) through the GPU's sigmoid too (it may differ from the CPU's by an ulp):
    # pick, among the neighbouring fp32 logits, one for which the reference's CPU sigmoid and the HIP sigmoid both do
    cands = [np.float32(g["l04"])]
    for step in (1.0, -1.0):
        v = np.float32(g["l04"])
        for _ in range(12):
            v = np.nextafter(v, np.float32(step)); cands.append(v)
    cands = np.array(sorted(cands), np.float32)
    both = (O.clamped_sigmoid(cands) == c32) & (clamped_sigmoid(dev(cands)).cpu().numpy() == c32)
    assert both.any(), "no logit maps to fp32(0.4) under both sigmoids"
    l04 = cands[both][both.sum() // 2]
    for (c, x, y) in g["planted_cells"]:
        assert head[0, c, y, x] == g["l04"]
        head[0, c, y, x] = l04
    args = make_args(M, N, K, P)
    dec = Decoder(args)
    md = dec(head_views(dev(head), M, N), conf_thresh=conf, dist_thresh=dist, return_metadata=True)
    for grp, key in (("anchor", "topk_anchor"), ("part", "topk_kp")):
        s, i, c, y, x = (t.cpu().numpy() for t in md[key])
        ref_s = g[f"dec_{grp}_score"]
        pos = ref_s > 0
        np.testing.assert_array_equal(s > 0, pos)                                   # same masked / unmasked split (fp32 `>`)
        np.testing.assert_array_equal(i[pos], g[f"dec_{grp}_ind"][pos])
        np.testing.assert_array_equal(x[pos], g[f"dec_{grp}_x"][pos]); np.testing.assert_array_equal(y[pos], g[f"dec_{grp}_y"][pos])
    fast = dec(head_views(dev(head), M, N), conf_thresh=conf, dist_thresh=dist)
    for ann in (md["annotation"][0], fast[0]):
        o, p = annotation_arrays(args, ann)
        ro, rp = g["ann0_objs"], g["ann0_parts"]
        assert o.shape == ro.shape and p.shape == rp.shape
        np.testing.assert_array_equal(o[:, :3], ro[:, :3]); np.testing.assert_allclose(o[:, 3], ro[:, 3], **SIG_TOL)
        np.testing.assert_array_equal(p[:, :4], rp[:, :4])
        on_edge = o[:, 3] == float(c32)
        assert on_edge.sum() == 1 and not (p[:, 0] == np.nonzero(on_edge)[0][0]).any()   # emitted, and without parts
    r = np.array([[args.parts[k.kind], k.x, k.y, k.score] for k in md["raw_parts"][0]], np.float64).reshape(-1, 4)
    assert r.shape == g["raw0"].shape
    np.testing.assert_array_equal(r[:, :3], g["raw0"][:, :3])
    assert (r[:, 3] == float(c32)).sum() == 1


# ------------------------------------------------------------------------------------------ map-parallel decoder
def test_sigmoid_properties_behind_the_logit_domain_nms_hold_for_every_fp32_value():
    """k_nms_slots_v decides the 5x5 NMS on LOGITS (survivor <=> sigma(x) == sigma(window maximum)), which equals the reference's compare
    of clamped sigmoids (utils.py:441-443 after :355-361) iff the device's clamped sigmoid is monotone non-decreasing, and its near-tie
    margin table is valid iff a logit further below the maximum than the margin always has a strictly smaller sigmoid.  Both are
    checked here for ALL 2^32 bit patterns on the device."""
    from structuredetector_amd import _lib as L
    out = torch.zeros(3, dtype=torch.int64, device=DEV)
    L.check(L.lib().sd_selfcheck_sigmoid(out.data_ptr(), L.stream()))
    mono, margin, seen = out.cpu().tolist()
    assert seen == 2 ** 32 - 2 * (2 ** 23 - 1)                       # every non-NaN value
    assert mono == 0, f"{mono} consecutive fp32 pairs where the clamped sigmoid decreases"
    assert margin == 0, f"{margin} window maxima whose margin admits a tie"


@pytest.mark.parametrize("img,kind", [(512, "noise"), (512, "ties"), (1024, "noise"), (208, "ties"), (512, "planted"), (1024, "planted")])
def test_early_score_cut_of_the_streaming_decoder_over_thresholds(img, kind):
    """The streaming kernel of the map-parallel path drops a lane's pixels early when their logits are below a CONSERVATIVE logit of the score
    threshold (sd_decode.hip, conservative_min_logit: the exact `clamped_sigmoid(x) >= fp32(conf)` test of decoders.py:78,83 / utils.py:355-361
    still runs on what passes).  Swept over thresholds at both ends of the clamp, just around representable sigmoid values and beyond 0.999
    (where the cut is capped): the annotations-only result must equal the launch pair's, which has no such cut, bit for bit."""
    from structuredetector_amd import _lib as L
    from structuredetector_amd.data import Decoder
    B, M, N, K, P = 3, 2, 2, 24, 48
    rng = np.random.default_rng(img + len(kind))
    h = img // 4
    planted = None
    if kind == "noise":
        head = (4 * rng.standard_normal((B, M + N + 4, h, h))).astype(np.float32)                # sigmoids from ~1e-7 to ~1 - 1e-7
    elif kind == "planted":
        # isolated peaks on a flat floor, fewer than the lists hold, and thresholds placed ON their scores (and one ulp either side): a cut
        # that is not conservative -- the threshold's logit, or a hair above it -- loses exactly these (checked with such a build: this
        # case fails with `t + 2e-3`, the noise cases do not: there the peaks next to the threshold never make the lists)
        K, P = 64, 64
        head = np.full((B, M + N + 4, h, h), -20.0, np.float32)
        planted = np.concatenate([np.linspace(-13.5, 13.5, 14), rng.uniform(-14, 14, 6)]).astype(np.float32)
        for c in range(M + N):
            for i, v in enumerate(rng.permutation(planted)):
                head[:, c, 6 + 8 * (i // 5), 5 + 9 * (i % 5) + c] = v
    else:
        head = (np.round(3 * rng.standard_normal((B, M + N + 4, h, h))) * np.float32(2.5)).astype(np.float32)
        head[:, 0, :8, :] = 14.5; head[:, 0, 8:12, :] = np.float32(-13.9); head[:, 1, 4:6, :] = np.float32(6.9)      # clamp plateaus, sigmoid(6.9) = 0.99899
    views = head_views(dev(head), M, N)
    dec = Decoder(make_args(M, N, K, P))
    lib = L.lib()
    sig = lambda x: float(1.0 / (1.0 + np.exp(-np.float64(x))))
    confs = [0.0, 1e-7, 1e-6, 2e-6, 9e-6, 1.1e-5, 1e-4, 0.05, 0.4, 0.5, float(np.float32(sig(2.5))), float(np.nextafter(np.float32(sig(2.5)), np.float32(1))),
             0.9, 0.99, 0.999, 0.9990001, 0.9995, float(np.float32(sig(7.5))), 0.999999, 1.0]
    if planted is not None:
        for v in planted:
            s32 = np.float32(sig(v))
            confs += [float(s32), float(np.nextafter(s32, np.float32(0))), float(np.nextafter(s32, np.float32(1))), float(np.float32(sig(v - np.float32(1e-3))))]
    try:
        for conf in confs:
            L.check(lib.sd_decode_set_option(b"map_parallel_from", 1 << 30))
            want, _ = dec.decode_packed(views, conf, 0.1, exact_topk=False, fused=False)
            L.check(lib.sd_decode_set_option(b"map_parallel_from", 1))
            for split in (0, 1, 3):
                L.check(lib.sd_decode_set_option(b"map_split", split))
                got, _ = dec.decode_packed(views, conf, 0.1, exact_topk=False, fused=False)
                assert torch.equal(got, want), f"conf={conf!r} map_split={split}"
    finally:
        L.check(lib.sd_decode_set_option(b"map_parallel_from", -1))
        L.check(lib.sd_decode_set_option(b"map_split", 0))


@pytest.mark.parametrize("B,img,M,N,K,P,kind", [(4, 1024, 8, 8, 128, 512, "scene"), (2, 1024, 8, 8, 128, 512, "noise"), (64, 512, 2, 1, 20, 40, "scene"),
                                                (3, 264, 3, 2, 12, 24, "noise"), (2, 132, 1, 1, 3, 2, "flat"), (2, 512, 1, 2, 900, 1000, "noise"),
                                                (3, 512, 2, 2, 64, 200, "ties"), (2, 528, 2, 1, 20, 40, "ties"), (2, 266, 2, 1, 20, 40, "noise"),
                                                (3, 208, 2, 1, 20, 40, "noise"), (2, 400, 1, 2, 30, 50, "ties"), (130, 512, 2, 1, 20, 40, "scene"),
                                                (5, (384, 512), 2, 1, 20, 40, "noise"), (3, (512, 320), 1, 2, 20, 40, "ties"), (2, (448, 1024), 2, 2, 64, 96, "noise")])
def test_map_parallel_decoder_is_bit_identical_to_two_launch_path(B, img, M, N, K, P, kind):
    """sd_decode's map-parallel path (tile pass without global atomics -> one selector block per MAP = the reference's per-class top-k,
    utils.py:451 -> one merge + association block per image = its second top-k, utils.py:459) against the launch pair with one selector
    block per image: identical packed buffers bit for bit, both selection modes, both tile heights; lists that fit the ranking sort,
    the LDS radix select, the global radix select (one plateau: 64 k candidates in one map) and K / P close to the 1024 limit."""
    from structuredetector_amd import _lib as L
    from structuredetector_amd.data import Decoder
    ih, iw = img if isinstance(img, tuple) else (img, img)      # (non-square maps: bands go by the height, strips and the two-bands-per-wave form by the width)
    rng = np.random.default_rng(B * 131 + ih + K)
    h, w = ih // 4, iw // 4
    if kind == "noise":
        head = (2 * rng.standard_normal((B, M + N + 4, h, w))).astype(np.float32)
    elif kind == "flat":
        head = np.full((B, M + N + 4, h, w), -20.0, np.float32)
        head[:, :, 5, 7] = 3.0
    elif kind == "ties":
        # what the logit-domain tile pass must get right: exact ties (coarse grid), near-ties one ulp apart, both saturated ends of the
        # clamp (|x| > 13.8: plateaus where every tied pixel survives), windows whose maximum sits in each band of the margin table
        q = np.round(3 * rng.standard_normal((B, M + N + 4, h, w))) * np.float32(2.5)             # grid of 2.5: many equal neighbours, |x| up to ~30
        head = q.astype(np.float32)
        bump = rng.random(head.shape) < 0.3
        head[bump] = np.nextafter(head[bump], np.float32(np.inf))                                # one ulp above a neighbour's value
        head[:, 0, :8, :] = 14.5; head[:, 0, 8:16, :] = np.float32(13.9); head[:, 0, 16:20, :] = np.float32(-13.9)
        for i, v in enumerate((3.9, 4.0, 7.99, 8.0, 10.99, 11.0, 12.99, 13.0, 13.7, -13.69, -13.71)):
            head[:, 1, 30 + 2 * (i // 4), 10 + 8 * (i % 4)] = np.float32(v)
            head[:, 1, 30 + 2 * (i // 4), 11 + 8 * (i % 4)] = np.nextafter(np.float32(v), np.float32(-np.inf))
            head[:, 1, 30 + 2 * (i // 4), 12 + 8 * (i % 4)] = np.float32(v) - np.float32(0.003)
    else:
        n_max = 96 if K > 20 else 12
        head = np.stack([O.head_from_targets(rng, O.encode(img, img, O.synthetic_scene(rng, img, img, M, N, n_max // 2, n_max), M, N, K, P, 4.0, 0.1),
                                             M, N, noise=0.3) for _ in range(B)])
    if kind == "noise" and K == 128:
        head[0, 1] = -20.0                                                                       # one whole map a plateau: 65 536 tied candidates
    views = head_views(dev(head), M, N)
    dec = Decoder(make_args(M, N, K, P))
    lib = L.lib()
    try:
        for exact in (True, False):
            L.check(lib.sd_decode_set_option(b"map_parallel_from", 1 << 30))
            want, _ = dec.decode_packed(views, 0.5, 0.1, exact_topk=exact, fused=False)
            L.check(lib.sd_decode_set_option(b"map_parallel_from", 1))
            # streaming kernel (tile pass + per-map selection in one) / logit-domain tile kernel / per-pixel-sigmoid tile kernel + k_select_map
            # round 5: every map split over 1 .. 4 blocks of the streaming kernel (split 0 = the rule by geometry), and ranks + association in
            # ONE launch (k_rank_group, rank_group = 1) or as k_rank_maps + k_group_wide (0)
            # (rank_group 1 = by size: k_rank_group_small for K, P <= 64, the generic one-block kernel up to 1024 keys, the pair beyond; 2 = the
            # generic one-block kernel wherever its lists fit LDS)
            # half = 1 (default): on maps up to 128 columns wide the two halves of a wave walk two bands side by side; 0: one band per wave
            variants = [(0, 0, 1, 0, 1, 1), (0, 0, 1, 0, 0, 1), (0, 0, 1, 0, 2, 1), (0, 0, 1, 1, 1, 1), (0, 0, 1, 2, 1, 1), (0, 0, 1, 3, 0, 1), (0, 0, 1, 3, 2, 1),
                        (0, 0, 1, 4, 1, 1), (0, 0, 1, 0, 1, 0), (0, 0, 1, 1, 0, 0), (0, 0, 1, 2, 1, 0), (0, 0, 1, 3, 2, 0), (0, 0, 1, 4, 1, 0),
                        (16, 0, 0, 0, 1, 1), (32, 0, 0, 0, 0, 1), (16, 1, 0, 0, 2, 1), (32, 1, 0, 0, 0, 1)]
            for th, scalar, stream, split, rank_group, half in variants:
                L.check(lib.sd_decode_set_option(b"map_tile_height", th))
                L.check(lib.sd_decode_set_option(b"map_scalar_nms", scalar))
                L.check(lib.sd_decode_set_option(b"map_stream", stream))
                L.check(lib.sd_decode_set_option(b"map_split", split))
                L.check(lib.sd_decode_set_option(b"map_rank_group", rank_group))
                L.check(lib.sd_decode_set_option(b"map_half", half))
                # parts of three wave-iterations (128-row maps in two parts with two bands per wave, in four with one) on 192-thread blocks
                # (2 = always) and on 256-thread blocks with an idle wave (0)
                # bands of 128-row maps: 8 rows with two bands per wave on launches of < 1024 maps (1 = by size), 11 rows forced, 16 rows (0)
                for waves3, rows11 in (((2, 1), (0, 1), (2, 11)) if split in (2, 4) else ((1, 1), (1, 11), (1, 0)) if stream and split < 2 else ((1, 1),)):
                    L.check(lib.sd_decode_set_option(b"map_waves3", waves3))
                    L.check(lib.sd_decode_set_option(b"map_rows11", rows11))
                    for _ in range(2):                                                            # (back to back: no state left behind)
                        got, _ = dec.decode_packed(views, 0.5, 0.1, exact_topk=exact, fused=False)
                        assert torch.equal(got, want), (f"exact_topk={exact} map_tile_height={th} map_scalar_nms={scalar} map_stream={stream} map_split={split} "
                                                        f"map_rank_group={rank_group} map_half={half} map_waves3={waves3} map_rows11={rows11}")
    finally:
        L.check(lib.sd_decode_set_option(b"map_parallel_from", -1))
        L.check(lib.sd_decode_set_option(b"map_tile_height", 0))
        L.check(lib.sd_decode_set_option(b"map_scalar_nms", 0))
        L.check(lib.sd_decode_set_option(b"map_stream", 1))
        L.check(lib.sd_decode_set_option(b"map_split", 0))
        L.check(lib.sd_decode_set_option(b"map_rank_group", 1))
        L.check(lib.sd_decode_set_option(b"map_half", 1))
        L.check(lib.sd_decode_set_option(b"map_waves3", 1))
        L.check(lib.sd_decode_set_option(b"map_rows11", 1))
    if kind == "scene":
        t = O.decode_tensors(head[:, :M], head[:, M:M + N], head[:, M + N:M + N + 2], head[:, M + N + 2:], K, P, 0.5, 0.1)
        L.check(lib.sd_decode_set_option(b"map_parallel_from", 1))
        try:
            got, _ = dec.decode_packed(views, 0.5, 0.1, exact_topk=True, fused=False)
        finally:
            L.check(lib.sd_decode_set_option(b"map_parallel_from", -1))
        assert_decode_matches_oracle(dec.split_packed(got.cpu().numpy(), B, K, P), t, 0.5, SIG_TOL)


# ------------------------------------------------------------------------------------------ one-launch decoder
@pytest.mark.parametrize("B,img,M,N,K,P,kind", [(1, 512, 2, 1, 20, 40, "scene"), (64, 512, 2, 1, 20, 40, "scene"), (5, 256, 3, 2, 12, 24, "noise"),
                                                (3, 1024, 8, 8, 128, 512, "scene"), (2, 512, 8, 8, 128, 512, "noise"), (2, 132, 1, 1, 3, 2, "flat")])
def test_fused_decoder_is_bit_identical_to_two_launch_path(B, img, M, N, K, P, kind):
    """sd_decode_fused (one launch; the last tile block of an image selects and associates; self-cleaning counters) against
    sd_decode (NMS launch + select launch + memset): the packed result must be identical bit for bit, in both selection modes,
    on repeated calls (state left zero), for lists that take the LDS sort and for lists that take the radix select."""
    from structuredetector_amd.data import Decoder
    rng = np.random.default_rng(B * 31 + img + K)
    h = img // 4
    if kind == "noise":
        head = (2 * rng.standard_normal((B, M + N + 4, h, h))).astype(np.float32)               # ~4 % of all pixels are candidates
    elif kind == "flat":
        head = np.full((B, M + N + 4, h, h), -20.0, np.float32)                                 # one plateau: every pixel survives, all tied
        head[:, :, 5, 7] = 3.0
    else:
        n_max = 96 if K > 20 else 12
        head = np.stack([O.head_from_targets(rng, O.encode(img, img, O.synthetic_scene(rng, img, img, M, N, n_max // 2, n_max), M, N, K, P, 4.0, 0.1),
                                             M, N, noise=0.3) for _ in range(B)])
    views = head_views(dev(head), M, N)
    dec = Decoder(make_args(M, N, K, P))
    from structuredetector_amd import _lib as L
    try:
        for tall_from in (1 << 30, 1, 2688):                                                     # 64x16 tiles, 64x32 tiles, the default switch
            L.check(L.lib().sd_decode_set_option(b"tall_tiles_from", tall_from))
            for exact in (True, False, True):
                want, _ = dec.decode_packed(views, 0.5, 0.1, exact_topk=exact, fused=False)
                for _ in range(3):                                                               # back-to-back: no memset in between
                    got, _ = dec.decode_packed(views, 0.5, 0.1, exact_topk=exact, fused=True)
                    assert torch.equal(got, want), f"exact_topk={exact} tall_tiles_from={tall_from}"
    finally:
        L.check(L.lib().sd_decode_set_option(b"tall_tiles_from", 2688))
    state = next(iter(dec._state.values()))
    assert int(state.view(torch.int32).abs().sum()) == 0                                        # counters left zero
    if kind == "scene":
        t = O.decode_tensors(head[:, :M], head[:, M:M + N], head[:, M + N:M + N + 2], head[:, M + N + 2:], K, P, 0.5, 0.1)
        got, _ = dec.decode_packed(views, 0.5, 0.1, exact_topk=True)                             # default path = fused
        assert_decode_matches_oracle(dec.split_packed(got.cpu().numpy(), B, K, P), t, 0.5, SIG_TOL)


def test_fused_decoder_limits_and_fallback():
    from structuredetector_amd import _lib as L
    from structuredetector_amd.data import Decoder
    head = dev((2 * np.random.default_rng(0).standard_normal((1, 6, 64, 64))).astype(np.float32))
    assert L.lib().sd_decode_fused_supported(1, 2, 1, 128, 128, 20, 40) == 1 and L.lib().sd_decode_fused_supported(300, 2, 1, 128, 128, 20, 40) == 0
    dec = Decoder(make_args(1, 1, 600, 700))                      # beyond sd_decode_fused's 512: the two-launch path is taken
    a, _ = dec.decode_packed(head_views(head, 1, 1), 0.5, 0.1)
    b, _ = dec.decode_packed(head_views(head, 1, 1), 0.5, 0.1, fused=False)
    assert torch.equal(a, b) and not dec._state
    with pytest.raises(L.SdError, match="out of range"):
        dec.decode_packed(head_views(head, 1, 1), 0.5, 0.1, fused=True)


def test_queued_decodes_own_their_hand_off_state():
    """`Decoder.submit` keeps decodes in flight whose status nobody has read yet.  sd_decode_fused's hand-off records are self-validating, so a
    record left behind by a timed-out call would be accepted by the NEXT launch on the same state buffer: every queued submission therefore
    runs on a state buffer of its own, and a failed one is re-zeroed before it is reused.  Checked: two submissions in flight hold distinct
    buffers; a submission whose status word is made non-zero (the GPU cannot be made to time out on demand) goes through the two-launch
    redo, its -- deliberately dirtied -- buffer comes back ZERO, the other submission's result is untouched, and both results equal the
    synchronous call's."""
    from structuredetector_amd.data import Decoder
    M, N, K, P = 2, 1, 20, 40
    rng = np.random.default_rng(3)
    args = make_args(M, N, K, P, device=torch.device(DEV))
    scenes = [O.synthetic_scene(rng, 512, 512, M, N, 4, 9) for _ in range(8)]
    enc = [O.encode(512, 512, s, M, N, K, P, 4.0, 0.1) for s in scenes]
    heads = [dev(np.stack([O.head_from_targets(rng, e, M, N, noise=0.05) for e in enc[i:i + 4]])) for i in (0, 4)]
    dec = Decoder(args)
    want = [dec(head_views(h, M, N)) for h in heads]
    first = dec.submit(head_views(heads[0], M, N))
    second = dec.submit(head_views(heads[1], M, N))
    assert first.state is not None and second.state is not None, "bs = 4 at the cfg shape is served by the one-launch kernel"
    assert first.state.data_ptr() != second.state.data_ptr()
    second_ptr = second.state.data_ptr()
    torch.cuda.synchronize()
    assert int(first.state.count_nonzero()) == 0 and int(second.state.count_nonzero()) == 0      # left zero by a successful call
    first.state[100:164] = 0xA5                                         # what a late tile block of a timed-out call leaves behind
    first.host.numpy()[-4] = 1                                          # status word of image 0: "selector gave up"
    dirty = first.state
    got0, _ = first.result()
    assert dec.selector_timeouts == 1
    assert int(dirty.count_nonzero()) == 0, "the failed submission's state buffer must be re-zeroed before it is reused"
    got1, _ = second.result()
    assert dec.selector_timeouts == 1
    for got, exp in ((got0, want[0]), (got1, want[1])):
        assert [repr(a) for a in got] == [repr(a) for a in exp]
    third = dec.submit(head_views(heads[0], M, N))                      # the ring is reused, not grown
    assert third.state.data_ptr() in (dirty.data_ptr(), second_ptr)
    got2, _ = third.result()
    assert [repr(a) for a in got2] == [repr(a) for a in want[0]]
    assert sum(len(v) for v in dec._state_free.values()) == 2


# ------------------------------------------------------------------------------------------ empty and ragged inputs
@pytest.mark.parametrize("hm_fn", ["mse", "focal"])
def test_empty_and_ragged_batch_encode_loss_decode_vs_oracle(hm_fn):
    """One batch with every ragged case the reference's collate can produce (dataset.py:58-87 stacks fixed-size fields whatever the
    annotation holds): an image with NO objects, an object with NO parts, objects whose parts are all outside the image (clipped to the
    border, utils.py:364-381), and an ordinary scene.  Encode vs the oracle (indices / masks bit-exact), the loss on a seeded head
    (value and gradient; the empty image contributes no L1 terms), the decoder on the rendered targets (the empty image decodes to NO
    objects) and on a head that is below the threshold everywhere (all annotations empty)."""
    from structuredetector_amd.data import Decoder, Encode
    from structuredetector_amd.model import Loss
    W = H = 128
    M, N, K, P = 2, 2, 6, 9
    args = make_args(M, N, K, P, device=torch.device(DEV), hm_loss_fn=hm_fn)
    scenes = [
        [],                                                                                   # no objects at all
        [(0, 40.25, 70.5, [])],                                                               # an anchor without parts
        [(1, 20.0, 20.0, [(0, -15.0, 300.0), (1, 500.0, -3.0)]), (0, 127.9, 0.0, [])],        # parts outside the image: clipped
        [(0, 30.5, 31.5, [(0, 50.0, 60.0), (1, 12.0, 90.75)]), (1, 99.0, 64.0, [(1, 101.5, 80.0)])],
    ]
    anns = [to_annotation(args, s, f"img{i}.png") for i, s in enumerate(scenes)]
    out = Encode(args).batch((W, H), anns)
    want = O.collate([O.encode(W, H, s, M, N, K, P, 4.0, 0.1) for s in scenes])
    for k in ENC_KEYS:
        got = out[k].cpu().numpy()
        assert got.shape == want[k].shape and got.dtype == want[k].dtype, k
        if k.endswith("_hm"):
            np.testing.assert_allclose(got, want[k], rtol=1e-5, atol=1e-7, err_msg=k)
            np.testing.assert_array_equal(got == 1.0, want[k] == 1.0)
        else:
            np.testing.assert_array_equal(got, want[k], err_msg=k)
    assert not out["anchor_mask"][0].any() and not out["part_mask"][0].any() and float(out["anchor_hm"][0].abs().sum()) == 0.0
    assert int(out["anchor_mask"][1].sum()) == 1 and not out["part_mask"][1].any()

    # loss: value + gradient vs the oracle on a seeded head
    rng = np.random.default_rng(11)
    head = rng.standard_normal((4, M + N + 4, H // 4, W // 4)).astype(np.float32)
    ref = O.loss(head, want, M, N, hm_loss_fn=hm_fn, want_grad=True)
    hd = dev(head).requires_grad_(True)
    crit = Loss(args)
    val = crit(head_views(hd * 1.0, M, N), out)
    val.backward()
    np.testing.assert_allclose([val.item(), float(crit.stats.hm_loss), float(crit.stats.offset_loss), float(crit.stats.embedding_loss)],
                               [ref["total"], ref["hm"], ref["offset"], ref["embedding"]], rtol=1e-4, atol=1e-7)
    np.testing.assert_allclose(hd.grad.cpu().numpy(), ref["grad"], rtol=1e-4, atol=1e-4 * np.abs(ref["grad"]).max())

    # decoder on heads synthesised from these targets: the empty image yields no objects, the part-less anchor an object without parts
    heads = np.stack([O.head_from_targets(rng, {k: v[b] for k, v in want.items()}, M, N, noise=0.0) for b in range(4)])
    dec = Decoder(args)
    got_anns = dec(head_views(dev(heads), M, N))
    t = O.decode_tensors(heads[:, :M], heads[:, M:M + N], heads[:, M + N:M + N + 2], heads[:, M + N + 2:], K, P, args.conf_threshold,
                         args.decoder_dist_thresh)
    for b in range(4):
        o, p = annotation_arrays(args, got_anns[b])
        ro, rp = objects_to_arrays(O.assemble_objects(t, b, args.conf_threshold, 4.0, W // 4, H // 4))
        assert o.shape == ro.shape and p.shape == rp.shape, (b, o.shape, ro.shape, p.shape, rp.shape)
        np.testing.assert_array_equal(o[:, :3], ro[:, :3]); np.testing.assert_array_equal(p[:, :4], rp[:, :4])
    assert len(got_anns[0].objects) == 0
    assert len(got_anns[1].objects) == 1 and len(got_anns[1].objects[0].parts) == 0
    # a head below the threshold everywhere (and one of all zeros: sigmoid 0.5 is not > 0.5... conf_threshold decides) -> empty annotations
    low = dev(np.full((2, M + N + 4, H // 4, W // 4), -8.0, np.float32))
    assert all(len(a.objects) == 0 for a in dec(head_views(low, M, N)))
