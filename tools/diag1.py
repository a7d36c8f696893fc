import sys, importlib
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import numpy as np, torch
gs = importlib.import_module('pytorch-graphsage_amd')
ops = gs.ops
ops.warmup(torch.device('cuda'))
print('mm_out_f32', ops.config.mm_out_f32, gs._native.device_info())
torch.manual_seed(0)
for (M, h, D) in [(17, 128, 16), (70, 128, 602), (17, 128, 24), (33, 8, 8), (512, 128, 256)]:
    g = torch.randn(M, 2 * h, device='cuda').bfloat16()
    x = torch.randn(M, D, device='cuda').bfloat16()
    gx = g[:, :h]
    ref = gx.float().t() @ x.float()
    a = torch.mm(gx.t(), x, out_dtype=torch.float32)
    b = torch.mm(gx.t().contiguous(), x, out_dtype=torch.float32)
    c = torch.mm(gx.t(), x).float()
    print(M, h, D, 'strided', float((a - ref).abs().max()), 'contig', float((b - ref).abs().max()),
          'bf16out', float((c - ref).abs().max()), 'scale', float(ref.abs().max()))
