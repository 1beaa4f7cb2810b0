import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch, torch.nn.functional as F
from mj_video_amd import ops
from test_kernels_gpu import rnd
from util import bf16_ulps
BF = torch.bfloat16; cuda = torch.device("cuda")
M, K = 2200, 2048
x = rnd(M, K, std=1.5, seed=1); gain = (rnd(K, std=0.2, seed=2).float() + 1.0).to(BF); xc = x.to(cuda)
rstd = torch.empty(ops.padded_rows(M), dtype=torch.float32, device=cuda); ops.row_stats(xc, rstd, None, 1e-5)
xd = x.double(); r = 1.0 / torch.sqrt((xd * xd).mean(1, keepdim=True) + 1e-5)
ff = 256
w1, w3 = rnd(ff, K, std=0.1, seed=6), rnd(ff, K, std=0.1, seed=7)
w13 = torch.stack([w1.view(ff // 16, 16, K), w3.view(ff // 16, 16, K)], dim=1).reshape(2 * ff, K)
w13f = (w13.float() * gain.float()[None, :]).to(BF)
w1f, w3f = (w1.float() * gain.float()[None, :]).to(BF), (w3.float() * gain.float()[None, :]).to(BF)
gx = (r * (xd @ w1f.double().t())); ux = (r * (xd @ w3f.double().t()))
g = gx.float().to(BF); u = ux.float().to(BF)
ref = (F.silu(g.float()).to(BF).float() * u.float()).to(BF)
for tile in (64, 128, 256):
    out = torch.empty(M, ff, dtype=BF, device=cuda)
    ops.gemm(xc, w13f.to(cuda), out, ops.EPI_SILU_MUL, folded_norm=(rstd,), tile=tile)
    ul = bf16_ulps(out.float().cpu(), ref.float())
    bad = (ul > 3) & ((out.float().cpu() - ref.float()).abs() > 2e-3)
    print("tile", tile, "bad", int(bad.sum()))
    for (i, j) in bad.nonzero().tolist()[:5]:
        print("  at", i, j, "g exact", gx[i, j].item(), "g bf16", g[i, j].item(), "u", ux[i, j].item(), "out", out[i, j].item(), "ref", ref[i, j].item(),
              "silu(g) bf16", F.silu(g[i, j].float()).to(BF).item())
