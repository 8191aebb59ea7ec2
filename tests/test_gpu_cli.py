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
