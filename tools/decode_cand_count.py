import sys, numpy as np, torch
sys.path.insert(0, str(__import__('pathlib').Path(__file__).resolve().parent.parent))
from bench import make_args
from structuredetector_amd.data import Decoder, Encode
from structuredetector_amd.data.synthetic import synthetic_batch
dev = torch.device("cuda")
B=16; M, N, K, P, img = 8, 8, 128, 512, 1024
args = make_args(dev, M, N, K, P)
enc = Encode(args)
gen = torch.Generator(device=dev).manual_seed(0)
tgt = enc.render(enc.plan(img, img, *synthetic_batch(np.random.default_rng(B), B, img, img, M, N, 64, 96)), dev)
hm = torch.cat([tgt["anchor_hm"], tgt["part_hm"]], 1).clamp(1e-4, 0.95)
x = torch.log(hm / (1 - hm)) + 0.05 * torch.randn(hm.shape, device=dev, generator=gen)
s = torch.sigmoid(x)
mp = torch.nn.functional.max_pool2d(s, 5, 1, 2)
for thr in (0.0, 0.01, 0.1, 0.5):
    keep = (s == mp) & (s >= thr)
    n = keep.flatten(2).sum(2)
    print(thr, "anchors n per map: min/med/max", n[:, :M].min().item(), n[:, :M].float().median().item(), n[:, :M].max().item(), " parts:", n[:, M:].min().item(), n[:, M:].float().median().item(), n[:, M:].max().item())
