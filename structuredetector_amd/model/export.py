"""Fused inference export (reference: src/sdnet/cli/convert_coreml.py:12-29, `RawDecoder` + `CoreMLModel`):
the network with the first decoder stage -- clamped sigmoid + 5x5 NMS of the heatmap channels -- inside the model, so that
the consumer only runs top-k + grouping (`FusedOutputDecoder`).  The reference traces this module into a CoreML package for
Apple hardware; here the same module runs on the HIP kernels, optionally as ONE hipGraph (forward + NMS), and `save()` writes a
self-describing checkpoint."""
from __future__ import annotations

import torch

from .. import _lib as L
from ..data.decoders import RawDecoder
from .network import Network


class FusedInferenceModel(torch.nn.Module):
    """convert_coreml.py:21-29 (`CoreMLModel`): forward(image) = RawDecoder(model(image)); output (B, M+N+4, H/R, W/R) with the
    M+N heatmap channels already squashed and suppressed."""

    FORMAT = "sdnet-fused-inference-v1"

    def __init__(self, model: Network, args) -> None:
        super().__init__()
        self.model = model
        self.model.raw_output = True
        self.nb_hms = len(args.labels) + len(args.parts)
        self.decoder = RawDecoder(nb_hms=self.nb_hms)
        self.meta = {"labels": dict(args.labels), "parts": dict(args.parts), "fpn_depth": args.fpn_depth,
                     "down_ratio": getattr(args, "down_ratio", 4.0)}

    def forward(self, image: torch.Tensor) -> torch.Tensor:
        return self.decoder(self.model(image))

    def split(self, out: torch.Tensor):
        """(B, M+N+4, h, w) -> the dict `FusedOutputDecoder` consumes (same keys as Network.forward, network.py:79-84)."""
        M, nb = len(self.meta["labels"]), self.nb_hms
        return {"anchor_hm": out[:, :M], "part_hm": out[:, M:nb], "offsets": out[:, nb:nb + 2], "embeddings": out[:, nb + 2:nb + 4]}

    def graphed(self, example: torch.Tensor):
        """Capture forward + sigmoid/NMS for `example`'s shape into one hipGraph: run(x) -> fused output (static buffer)."""
        if self.model.training:
            raise L.SdError("graphed() captures the inference forward: call .eval() first")
        static_in = example.detach().clone().contiguous().float()
        with torch.no_grad():
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                for _ in range(2):
                    self.forward(static_in)
            torch.cuda.current_stream().wait_stream(side)
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph, stream=side):
                static_out = self.forward(static_in)

        def run(x):
            static_in.copy_(x, non_blocking=True)
            graph.replay()
            return static_out

        run.graph, run.static_in, run.static_out = graph, static_in, static_out
        return run

    def save(self, path):
        torch.save({"format": self.FORMAT, "meta": self.meta, "state_dict": {k: v.detach().cpu().clone() for k, v in self.model.state_dict().items()}}, path)

    @classmethod
    def load(cls, path, device="cuda"):
        from argparse import Namespace
        blob = torch.load(path, map_location="cpu", weights_only=True)      # tensors, dicts, tuples, strings and numbers only
        if blob.get("format") != cls.FORMAT:
            raise L.SdError(f"{path}: not a {cls.FORMAT} file")
        args = Namespace(labels=blob["meta"]["labels"], parts=blob["meta"]["parts"], fpn_depth=blob["meta"]["fpn_depth"],
                         down_ratio=blob["meta"]["down_ratio"])
        net = Network(args, pretrained=False, raw_output=True)
        net.load_state_dict(blob["state_dict"])
        return cls(net.to(device).eval(), args)
