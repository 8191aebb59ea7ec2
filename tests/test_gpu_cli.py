"""`train` / `evaluate` entry points end to end on the GPU (tiny synthetic runs) + the stress-config shapes."""
import json
from argparse import Namespace

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def test_train_and_evaluate_cli(tmp_path, monkeypatch, capsys):
    from structuredetector_amd.cli import evaluate, train
    monkeypatch.chdir(tmp_path)
    (tmp_path / "labels.json").write_text(json.dumps({"labels": ["bean", "maize"], "parts": ["leaf"]}))
    common = ["-W", "128", "-H", "128", "-s", "stem", "--labels", str(tmp_path / "labels.json")]
    train.main(common + ["--synthetic", "8", "-b", "4", "-e", "2", "--steps", "3"])
    out = capsys.readouterr().out
    assert "epoch 0: total" in out and "validation (" in out
    ckpts = list((tmp_path / "trainings").glob("*/model_best_loss.pth"))
    assert len(ckpts) == 1
    sd = torch.load(ckpts[0], map_location="cpu")
    assert "adpater.0.weight" in sd and sd["head.conv.weight"].shape == (7, 128, 1, 1)
    ev = evaluate.main(common + ["--synthetic", "3", "-o", str(ckpts[0]), "--save_csv_eval", str(tmp_path / "kps.csv")])
    out = capsys.readouterr().out
    assert "Anchor Location" in out and "CSI" in out and "Classification" in out
    assert ev.anchor_eval.reduce().npos > 0
    assert (tmp_path / "kps.csv").read_text().startswith("bean,")


def test_training_reduces_loss():
    """A few Adam steps on one fixed synthetic batch must drive the loss down (fwd + loss + bwd + Adam are consistent)."""
    from structuredetector_amd.data import Encode
    from structuredetector_amd.data.synthetic import synthetic_batch
    from structuredetector_amd.model import Network
    from structuredetector_amd.model.trainer import TrainStep
    from tests.test_host_cpu import make_args
    dev = torch.device("cuda")
    args = make_args(2, 1, 20, 40, device=dev, learning_rate=1e-3)
    net = Network(args, pretrained=False).to(dev).train()
    step = TrainStep(net, args)
    enc = Encode(args)
    tgt = enc.render(enc.plan(128, 128, *synthetic_batch(np.random.default_rng(0), 4, 128, 128, 2, 1)), dev)
    x = torch.randn(4, 3, 128, 128, device=dev)
    losses = [float(step(x, tgt)[0]) for _ in range(12)]
    assert np.isfinite(losses).all() and losses[-1] < 0.7 * losses[0], losses


def test_stress_config_shapes():
    """BASELINE configs[4] geometry in fp32: 8 labels / 8 parts (20 head channels), K=128, P=512, dense scenes."""
    from structuredetector_amd.data import Decoder, Encode
    from structuredetector_amd.data.synthetic import synthetic_batch
    from structuredetector_amd.model import Network
    from tests.test_host_cpu import make_args
    dev = torch.device("cuda")
    args = make_args(8, 8, 128, 512, device=dev)
    net = Network(args, pretrained=False).to(dev).eval()
    x = torch.randn(1, 3, 256, 256, device=dev)
    with torch.no_grad():
        out = net(x)
    assert out["anchor_hm"].shape == (1, 8, 64, 64) and out["embeddings"].shape == (1, 2, 64, 64)
    enc = Encode(args)
    tgt = enc.render(enc.plan(256, 256, *synthetic_batch(np.random.default_rng(1), 1, 256, 256, 8, 8, 64, 96)), dev)
    assert int(tgt["anchor_mask"].sum()) >= 64
    anns = Decoder(args)(out)
    assert len(anns) == 1


@pytest.mark.parametrize("eval_batch", [16, 5, 1])
def test_evaluate_on_16_png_json_samples_vs_reference(golden_dir, tmp_path, monkeypatch, capsys, eval_batch):
    """BASELINE configs[0]: `evaluate` over a directory of 16 PNG + JSON samples (2 labels / 1 part, anchor_name=stem), read
    by the product CropDataset, decoded by the HIP decoder, scored by the product Evaluator -- against the reference's own
    from_json / Resize / Encode-clip / Decoder / Evaluator on the same head tensors (tests/golden/evaluate16.npz).  The head
    tensors are planted (a stand-in Network returns them in file order: a random-init backbone has nothing to detect), so the
    counters, accuracy lists and the CSV must match the reference EXACTLY; the real backbone then runs over the same directory
    for the plumbing, and its first head is checked against the oracle network on the image the reader produced.
    `evaluate` runs batched (decode threads -> GPU Resize + Normalize -> forward + decoder `--eval_batch` images per launch, next
    batch queued before this one is assembled): whole batches, a ragged last batch (16 = 3 x 5 + 1) and the reference's batch of one
    must all give the reference's numbers, and every image tensor the network is handed must equal, bit for bit, the host chain
    PIL resize -> to_tensor -> Normalize of the reference's ValidationAugmentation (transforms.py:255-261)."""
    from oracle import sdnet_oracle as O
    from structuredetector_amd.cli import evaluate
    from structuredetector_amd.data import CropDataset
    from structuredetector_amd.model import Network
    from tests.helpers import assert_evaluator_equals_golden, write_evaluate16_dir
    g = np.load(golden_dir / "evaluate16.npz")
    heads = write_evaluate16_dir(g, tmp_path / "valid")
    (tmp_path / "labels.json").write_text(json.dumps({"labels": ["bean", "maize"], "parts": ["leaf"]}))
    argv = ["--valid_dir", str(tmp_path / "valid"), "-s", "stem", "--labels", str(tmp_path / "labels.json"),
            "--save_csv_eval", str(tmp_path / "kps.csv")]
    batch_flag = ["--eval_batch", str(eval_batch)]
    seen = []

    class PlantedNetwork(torch.nn.Module):
        def __init__(self, args, *a, **kw):
            super().__init__()
            self.dummy = torch.nn.Parameter(torch.zeros(1))

        def forward(self, x):
            assert tuple(x.shape[1:]) == (3, 512, 512) and x.shape[0] <= eval_batch and x.is_cuda
            lo = len(seen)
            seen.extend(x.cpu())
            h = torch.from_numpy(np.stack(heads[lo:lo + x.shape[0]])).to(x.device)
            return {"anchor_hm": h[:, :2], "part_hm": h[:, 2:3], "offsets": h[:, 3:5], "embeddings": h[:, 5:7]}

    monkeypatch.setattr(evaluate, "Network", PlantedNetwork)
    ev = evaluate.main(argv + batch_flag)
    assert len(seen) == 16
    host_reader = CropDataset(evaluate.Arguments().parse(argv[:-2]), tmp_path / "valid")       # PIL resize + to_tensor + Normalize on the host
    for i in range(16):
        assert torch.equal(seen[i], host_reader[i][0]), f"image {i}: GPU Resize + Normalize differs from the PIL chain"
    assert_evaluator_equals_golden(ev, g)
    assert (tmp_path / "kps.csv").read_text() == str(g["csv"])
    out = capsys.readouterr().out
    assert "Anchor Location" in out and "CSI" in out
    monkeypatch.undo()

    # the real network over the same directory (seeded random checkpoint through --load_model)
    ref = O.build_reference_network(2, 1, seed=16)
    torch.save(ref.state_dict(), tmp_path / "seeded.pth")
    ev2 = evaluate.main(argv[:-2] + batch_flag + ["-o", str(tmp_path / "seeded.pth")])
    assert ev2.anchor_eval.reduce().npos == ev.anchor_eval.reduce().npos == 110
    args = evaluate.Arguments().parse(argv[:-2])
    image, _ = CropDataset(args, tmp_path / "valid")[1]                    # a 640x480 PNG resized to 512x512 and normalised
    net = Network(args, raw_output=True)
    net.load_state_dict(torch.load(tmp_path / "seeded.pth", map_location="cpu"))
    with torch.no_grad():
        got = net.eval().to("cuda")(image[None].to("cuda")).cpu()
        want = ref.eval()(image[None])
    assert (got - want).abs().max().item() <= 1e-4 * want.abs().max().item()


def test_resume_is_bit_identical(tmp_path, monkeypatch):
    """f3 true resume: 4 optimizer steps == 2 steps + save (weights, BatchNorm buffers, Adam moments + step, StepLR epoch) + load
    into a FRESH process-state + 2 steps, bit for bit (trainer.py:226-237 saves weights only)."""
    from structuredetector_amd.cli import train
    from structuredetector_amd.data import Encode
    from structuredetector_amd.data.synthetic import synthetic_batch
    from structuredetector_amd.model import Network
    from structuredetector_amd.model.trainer import StepLR, TrainStep
    from tests.test_host_cpu import make_args
    dev = torch.device("cuda")
    args = make_args(2, 1, 20, 40, device=dev, learning_rate=1e-3)
    enc = Encode(args)
    batches = []
    for i in range(4):
        tgt = enc.render(enc.plan(128, 128, *synthetic_batch(np.random.default_rng(10 + i), 4, 128, 128, 2, 1)), dev)
        batches.append((torch.randn(4, 3, 128, 128, device=dev, generator=torch.Generator(dev).manual_seed(20 + i)), tgt))

    def fresh():
        torch.manual_seed(3)
        net = Network(args, pretrained=False).to(dev).train()
        step = TrainStep(net, args)
        return net, step, StepLR(step, 1)

    net_a, step_a, sch_a = fresh()
    for i, (x, t) in enumerate(batches):
        step_a(x, t)
        if i == 1:
            sch_a.step()                                  # lr drops by 10x after the second step
    net_b, step_b, sch_b = fresh()
    for i, (x, t) in enumerate(batches[:2]):
        step_b(x, t)
    sch_b.step()
    torch.save({"model": net_b.state_dict(), "optimizer": step_b.state_dict(), "scheduler": sch_b.state_dict()}, tmp_path / "resume.pth")
    del net_b, step_b, sch_b
    state = torch.load(tmp_path / "resume.pth", map_location="cpu", weights_only=False)
    torch.manual_seed(99)                                 # a different init: everything must come from the file
    net_c = Network(args, pretrained=False).to(dev).train()
    net_c.load_state_dict(state["model"])
    step_c = TrainStep(net_c, args)
    sch_c = StepLR(step_c, 1)
    step_c.load_state_dict(state["optimizer"]); sch_c.load_state_dict(state["scheduler"])
    assert step_c.step_count == 2 and abs(step_c.lr - 1e-4) < 1e-12 and sch_c.epoch == 1
    for x, t in batches[2:]:
        step_c(x, t)
    assert torch.equal(net_c.flat_params, net_a.flat_params)
    assert torch.equal(step_c.exp_avg, step_a.exp_avg) and torch.equal(step_c.exp_avg_sq, step_a.exp_avg_sq)
    for (ka, va), (kc, vc) in zip(net_a.state_dict().items(), net_c.state_dict().items()):
        assert ka == kc and torch.equal(va, vc), ka        # running statistics and num_batches_tracked too
    bad = dict(state["optimizer"], flat_numel=123)
    from structuredetector_amd import _lib as L
    with pytest.raises(L.SdError):
        step_c.load_state_dict(bad)

    # Trainer-level: resume.pth written next to the checkpoints, --resume continues at the next epoch; --amp refused
    monkeypatch.chdir(tmp_path)
    (tmp_path / "labels.json").write_text(json.dumps({"labels": ["bean", "maize"], "parts": ["leaf"]}))
    common = ["-W", "128", "-H", "128", "-s", "stem", "--labels", str(tmp_path / "labels.json"), "--synthetic", "8", "-b", "4"]
    train.main(common + ["-e", "1"])
    resume = list((tmp_path / "trainings").glob("*/resume.pth"))
    assert len(resume) == 1
    st = torch.load(resume[0], map_location="cpu", weights_only=True)
    assert st["epoch"] == 0 and st["optimizer"]["step_count"] == 2 and st["scheduler"]["epoch"] == 1
    train.main(common + ["-e", "2", "--resume", str(resume[0])])


def test_resume_restores_augmentation_state_over_a_directory(golden_dir, tmp_path, monkeypatch):
    """`resume.pth` also carries the multi-scale size drawn for the next epoch, the random streams (torch's global generator,
    the rank's numpy generator) and the run's save directory, and loads with weights_only=True: a resumed run continues with the input
    size and the draws the uninterrupted run would have had, and keeps writing into the same trainings/<stamp>/."""
    from structuredetector_amd.model.trainer import Trainer
    from structuredetector_amd.utils.args import Arguments
    from tests.helpers import write_evaluate16_dir
    g = np.load(golden_dir / "evaluate16.npz")
    write_evaluate16_dir(g, tmp_path / "train")
    (tmp_path / "labels.json").write_text(json.dumps({"labels": ["bean", "maize"], "parts": ["leaf"]}))
    monkeypatch.chdir(tmp_path)
    argv = ["--train_dir", str(tmp_path / "train"), "-W", "128", "-H", "128", "-s", "stem", "--labels", str(tmp_path / "labels.json"), "-b", "8", "-e", "3"]
    args = Arguments().parse(argv)
    torch.manual_seed(77)
    tr = Trainer(args)
    sizes = []
    orig = type(tr.augment).trigger_random_resize

    def spy(self):
        sizes.append(orig(self))
        return sizes[-1]
    monkeypatch.setattr(type(tr.augment), "trigger_random_resize", spy)
    args.epochs = 1
    tr.train()                                              # epoch 0, then the size of epoch 1 is drawn and resume.pth written
    resume = tr.save_dir / "resume.pth"
    state = torch.load(resume, map_location="cpu", weights_only=True)
    assert tuple(state["augment_size"]) == tuple(sizes[-1]) == tuple(tr.augment.size)
    want_draw, want_np = torch.rand(3), tr.rng.random(3)    # what the uninterrupted run would draw next
    torch.manual_seed(1234)                                 # a fresh process would start from other streams
    args2 = Arguments().parse(argv + ["--resume", str(resume)])
    tr2 = Trainer(args2)
    assert tr2.start_epoch == 1 and tuple(tr2.augment.size) == tuple(sizes[-1]) and tr2.save_dir == tr.save_dir
    assert torch.equal(torch.rand(3), want_draw) and np.array_equal(tr2.rng.random(3), want_np)
    assert tr2.step.step_count == tr.step.step_count and torch.equal(tr2.net.flat_params, tr.net.flat_params)


def test_fused_inference_export_and_detect(tmp_path, monkeypatch):
    """f4: RawDecoder (convert_coreml.py:12-18) = cat(nms(clamped_sigmoid(hm)), regs) in one tile pass; FusedInferenceModel
    (network + that stage, also as one hipGraph) + FusedOutputDecoder give the same annotations as Network + Decoder;
    save / load round trip; `detect` over a folder of .jpg writes predictions/<name>.json + the drawn image."""
    from PIL import Image
    from oracle import sdnet_oracle as O
    from structuredetector_amd.cli import detect
    from structuredetector_amd.data import Decoder, FusedOutputDecoder, RawDecoder
    from structuredetector_amd.model import FusedInferenceModel, Network
    from structuredetector_amd.utils import ImageAnnotation, clamped_sigmoid, nms
    from tests.test_host_cpu import make_args
    dev = torch.device("cuda")
    rng = np.random.default_rng(8)
    M, N, K, P, img = 2, 1, 20, 40, 256
    args = make_args(M, N, K, P, device=dev)
    # RawDecoder against the (golden-pinned) primitives and the oracle, batch 1 and batch 3, views and copies
    for B in (1, 3):
        head = torch.from_numpy(np.stack([O.head_from_targets(rng, O.encode(img, img, O.synthetic_scene(rng, img, img, M, N), M, N, K, P, 4.0, 0.1),
                                                              M, N, noise=0.3) for _ in range(B)])).to(dev)
        raw = RawDecoder(M + N)(head)
        want = torch.cat([nms(clamped_sigmoid(head[:, :M + N])), head[:, M + N:]], 1)
        assert torch.equal(raw, want)
        ref = O.nms(O.clamped_sigmoid(head[:, :M + N].cpu().numpy()))
        np.testing.assert_array_equal(raw[:, :M + N].cpu().numpy() > 0, ref > 0)
        np.testing.assert_allclose(raw[:, :M + N].cpu().numpy(), ref, rtol=4e-7, atol=0)
        # decoding the fused output == decoding the logits
        v = {"anchor_hm": head[:, :M], "part_hm": head[:, M:M + N], "offsets": head[:, M + N:M + N + 2], "embeddings": head[:, M + N + 2:]}
        f = {"anchor_hm": raw[:, :M], "part_hm": raw[:, M:M + N], "offsets": raw[:, M + N:M + N + 2], "embeddings": raw[:, M + N + 2:]}
        a1, a2 = Decoder(args)(v), FusedOutputDecoder(args)(f)
        for x, y in zip(a1, a2):
            assert [(o.name, o.x, o.y, o.anchor.score, [(p.kind, p.x, p.y, p.score) for p in o.parts]) for o in x.objects] == \
                   [(o.name, o.x, o.y, o.anchor.score, [(p.kind, p.x, p.y, p.score) for p in o.parts]) for o in y.objects]
        assert sum(len(a) for a in a1) >= 3 * B
    # the exported module: network + sigmoid/NMS, eager and as one hipGraph, and through save / load
    net = Network(args, pretrained=False).to(dev).eval()
    fused = FusedInferenceModel(net, args)
    x = torch.randn(1, 3, 128, 160, device=dev)
    with torch.no_grad():
        out = fused(x)
        logits = net(x)
    assert out.shape == (1, M + N + 4, 32, 40)
    assert torch.equal(out, RawDecoder(M + N)(logits)) and torch.equal(out[:, M + N:], logits[:, M + N:])
    run = fused.graphed(x)
    assert torch.equal(run(x), out)
    fused.save(tmp_path / "fused.pt")
    again = FusedInferenceModel.load(tmp_path / "fused.pt")
    with torch.no_grad():
        assert torch.equal(again(x), out)
    assert set(fused.split(out)) == {"anchor_hm", "part_hm", "offsets", "embeddings"}
    # detect CLI
    monkeypatch.chdir(tmp_path)
    (tmp_path / "labels.json").write_text(json.dumps({"labels": ["bean", "maize"], "parts": ["leaf"]}))
    (tmp_path / "imgs").mkdir()
    for i, size in enumerate([(320, 240), (200, 200)]):
        Image.fromarray(rng.integers(0, 255, (size[1], size[0], 3), dtype=np.uint8)).save(tmp_path / "imgs" / f"p{i}.jpg")
    torch.save(O.build_reference_network(2, 1, seed=4).state_dict(), tmp_path / "w.pth")
    written = detect.main(["--valid_dir", str(tmp_path / "imgs"), "-W", "128", "-H", "128", "-s", "stem", "--labels", str(tmp_path / "labels.json"),
                           "-o", str(tmp_path / "w.pth"), "-t", "0.3"])
    assert [w.name for w in written] == ["p0.json", "p1.json"]
    ann = ImageAnnotation.from_json(tmp_path / "predictions" / "p0.json", "stem")
    assert list(ann.img_size) == [320, 240] and all(np.isfinite([o.x, o.y, o.anchor.score]).all() for o in ann.objects)
    assert Image.open(tmp_path / "predictions" / "p0.jpg").size == (320, 240)
