"""How many 32-row x 32-query blocks of the 10M-row scan could be decided by the bf16 hi*hi product alone?
(gap 1.)  A block needs the hi*lo / lo*hi corrections only if some entry comes within the hi*hi error
bound of its query's threshold.  Thresholds are per workgroup (256 workgroups, tiles interleaved): bracketed here by
the pre-pass threshold every list starts with (12th best of the first 32K rows: upper bound on the pass rate) and
the workgroup's final one (12th best of all its rows: lower bound)."""
import sys, torch
n, d, B, k = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000, 384, 128, 12
dev = "cuda"
g = torch.Generator(device=dev); g.manual_seed(4321)
q = torch.randn((B, d), generator=g, device=dev); q /= q.norm(dim=1, keepdim=True)
bound = 2.0 * (2 ** -9 + 2 ** -9)  # on 2*dot - |d|^2 for unit rows and queries
G, T = 256, 32
n_tiles = n // T
tiles_per_wg = n_tiles // G
n_use = tiles_per_wg * G * T
S = torch.empty((n_use, B), device=dev, dtype=torch.float32)
for c in range(0, n_use, 500_000):
    gg = torch.Generator(device=dev); gg.manual_seed(1234 + c // 500_000)
    x = torch.randn((min(500_000, n_use - c), d), generator=gg, device=dev); x /= x.norm(dim=1, keepdim=True)
    S[c:c + len(x)] = 2.0 * (x @ q.T) - 1.0
thr_sample = S[: 256 * 4 * T].topk(k, dim=0).values[-1]                      # [B]
Sv = S.view(tiles_per_wg, G, T, B)                                            # tile t of workgroup w = global tile t*G + w
blk_max = Sv.amax(dim=2).view(tiles_per_wg, G, B // 32, 32)                   # per (tile, wg, query): max over the 32 rows
per_wg = Sv.permute(1, 0, 2, 3).reshape(G, tiles_per_wg * T, B)
thr_final = torch.stack([per_wg[w].topk(k, dim=0).values[-1] for w in range(G)])  # [G, B]
def rate(thr):  # thr broadcastable to [tiles, G, B]
    need = (blk_max >= (thr - bound).view(*thr.shape[:-1], B // 32, 32)).any(dim=-1)   # any of the 32 queries of the block
    return need.float().mean().item()
up = rate(thr_sample.view(1, 1, B).expand(tiles_per_wg, G, B))
lo = rate(thr_final.view(1, G, B).expand(tiles_per_wg, G, B))
print(f"{n_use} rows, {B} queries, klist {k}: blocks needing the corrections: {lo*100:.1f} % (final per-workgroup thresholds) "
      f"to {up*100:.1f} % (pre-pass thresholds only) -> {1+2*lo:.2f} to {1+2*up:.2f} MFMAs per block instead of 3")
