import cProfile, pstats, sys, io, contextlib, time
from pathlib import Path
sys.path.insert(0, "/root/repo")
import torch
from tools.feed_bench import write_samples
from structuredetector_amd.cli import evaluate
root = Path("/tmp/sd_eval"); labels = write_samples(root / "valid", 256, 512)
argv = ["--valid_dir", str(root / "valid"), "--labels", str(labels), "-s", "stem"]
with contextlib.redirect_stdout(io.StringIO()):
    evaluate.main(argv)
torch.cuda.synchronize()
pr = cProfile.Profile(); t0 = time.perf_counter(); pr.enable()
with contextlib.redirect_stdout(io.StringIO()):
    evaluate.main(argv)
torch.cuda.synchronize(); pr.disable(); print("wall", time.perf_counter() - t0)
pstats.Stats(pr).sort_stats("cumulative").print_stats(35)
