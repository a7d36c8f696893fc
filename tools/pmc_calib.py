"""Known-byte-count launches for calibrating rocprofv3's FETCH_SIZE / WRITE_SIZE on gfx950
(MI355X_MICROARCH.md, HBM section: FETCH_SIZE reports half the bytes of a 16 B/lane coalesced stream,
WRITE_SIZE is uncalibrated -- "calibrate on a known byte count in your own access pattern").

Runs k_gather_mean as a plain row copy (n = 1) of M distinct rows of a bf16 table with 1280-byte rows:
every launch reads exactly M * 1280 B (+ 8 B ids per row) and writes exactly M * 1280 B with the
same 16 B/lane access shape as the step's gather kernels, far beyond the 256 MiB Infinity Cache.
tools/pmc_summary.py divides the known bytes by the raw counter of this kernel to get the factors it
applies to the step's kernels.   usage (under rocprofv3 --pmc ...): python tools/pmc_calib.py"""
import importlib
import sys

sys.path.insert(0, ".")
import torch

gs = importlib.import_module("pytorch-graphsage_amd")
M, LD = 400_000, 640                                   # 512 MB read + 512 MB written per launch
dev = torch.device("cuda")
table = torch.randn(M, LD, device=dev).bfloat16()
store = gs.FeatureStore(table, LD)
ids = torch.randperm(M, device=dev)
out = torch.empty_like(table)
for _ in range(6):
    gs.ops._gather_mean_raw(table, LD, ids, M, 1, torch.bfloat16, LD, out=out)
torch.cuda.synchronize()
print("calib rows=%d bytes_read=%d bytes_written=%d" % (M, M * LD * 2, M * LD * 2))
