"""Chained conv layers, the intermediate split image read back from the workspace: is a wrong output the producer's image or the
consumer's read?  (Found round 4's bug with it: in-flight asm loads into registers hipcc had already handed to the epilogue.)"""
import os, sys, torch
sys.path.insert(0, '/root/repo')
import aligner_amd
from aligner_amd.softattn import encode, _conv_workspaces
from aligner_amd import _lib
dev = torch.device("cuda:0")
FT = int(os.environ.get("FT", "0")); _lib.load().aligner_debug_set_option(b"conv_narrow_ft", FT); print("FT", FT)
g = torch.Generator().manual_seed(1)
def mk(ci, co, k): return (torch.randn(co, ci, k, generator=g).to(dev) / (ci*k)**0.5, (torch.randn(co, generator=g)*0.1).to(dev))
def up(v, a): return (v + a - 1) // a * a
for rep in range(12):
  for (B, T, ci, co) in [(3, 64, 160, 80), (3, 64, 160, 96), (8, 200, 160, 80)]:
    x = torch.randn(B, ci, T, generator=g).to(dev)
    l1 = mk(ci, co, 1)
    ident = (torch.eye(co).reshape(co, co, 1).to(dev).contiguous(), None)
    a = encode(x, [l1, ident]); torch.cuda.synchronize()
    ws = list(_conv_workspaces.bufs.values())[0]
    S = up(T, 16); nch1, nch2 = up(ci, 32) // 32, up(co, 32) // 32
    img = up((2 * B * max(nch1, nch2) * 4 * S + 512) * 16, 256)
    plane = B * nch2 * 4 * S * 16
    raw = ws[img: img + 2 * plane].clone()
    im = raw.view(torch.bfloat16).view(2, B, nch2, 4, S, 8).float()          # [plane][b][chunk][q][slot][j]
    got = (im[0] + im[1]).permute(0, 1, 2, 4, 3).reshape(B, nch2 * 32, S)[:, :co, :T]   # [b][chunk][q][j][slot] -> channels
    b = torch.relu(aligner_amd.conv1d(x, l1[0], l1[1], relu=False)); torch.cuda.synchronize()
    d_img = (got - b).abs(); d_out = (a - b).abs()
    bi = (d_img > 1e-3).nonzero(); bo = (d_out > 1e-4).nonzero()
    print(rep, B, T, ci, co, "image max err", float(d_img.max()), "nbad", len(bi), "| output max err", float(d_out.max()), "nbad", len(bo))
    for nm, bad in (("image", bi), ("output", bo)):
        if len(bad):
            print("   ", nm, "batches", sorted(set(bad[:,0].tolist()))[:10], "channels", sorted(set(bad[:,1].tolist()))[:40], "frames", sorted(set(bad[:,2].tolist()))[:40])
