import sys, numpy as np, torch
sys.path.insert(0, '.')
import aligner_amd
from oracle import forward_sum_oracle as FS
dev = torch.device("cuda:0")
def rl(rng, B, Tx, Ty, scale=3.0):
    z = rng.standard_normal((B, Tx, Ty)) * scale
    return (z - np.log(np.exp(z).sum(axis=1, keepdims=True))).astype(np.float32)
for (B, Tx, Ty, ragged) in [(3, 7, 19, True), (4, 64, 200, True), (2, 200, 1000, False), (2, 255, 600, True), (2, 256, 500, True), (1, 400, 900, True), (2, 33, 33, False), (1, 600, 700, True)]:
  for blank in (-1.0, -6.0):
    rng = np.random.default_rng(B * 777 + Tx)
    x = rl(rng, B, Tx, Ty)
    if ragged:
        ty = rng.integers(max(Tx // 2, 2), Ty + 1, size=B); tx = np.minimum(rng.integers(1, Tx + 1, size=B), ty); tx[0], ty[0] = Tx, Ty
    else:
        tx, ty = np.full(B, Tx), np.full(B, Ty)
    wl, wg = FS.ctc_forward_sum(x, tx, ty, blank)
    l, g = aligner_amd.forward_sum(torch.from_numpy(x).to(dev), torch.from_numpy(tx), torch.from_numpy(ty), blank_logprob=blank)
    l = l.cpu().numpy().astype(np.float64); g = g.cpu().numpy().astype(np.float64)
    worst = 0
    for b in range(B):
        K, T = int(tx[b]), int(ty[b])
        z = np.concatenate([np.full((1, T), blank), x[b, :K, :T].astype(np.float64)], 0)
        p = np.exp(z - np.log(np.exp(z).sum(0, keepdims=True)))[1:]
        occ = p - wg[b, :K, :T]
        d = np.abs(g[b, :K, :T] - wg[b, :K, :T])
        worst = max(worst, float((np.maximum(d - 2e-5, 0) / np.maximum(occ, 1e-12)).max()))
    print(B, Tx, Ty, blank, "loss err", np.abs(l - wl).max(), "rel", (np.abs(l - wl) / np.abs(wl)).max(), "worst (|d|-2e-5)/occ", worst)
