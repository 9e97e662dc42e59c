"""Cross-shard merge (mir_topk_merge_device) alone, as it runs after the all-gather at N GPUs: s shards x B queries
x k candidates in gathered blobs; checks against a numpy merge and times the kernel with HIP events."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from aidial_rag_amd import _native as nat

s, B, k = int(sys.argv[1]) if len(sys.argv) > 1 else 8, 96, 10
rng = np.random.default_rng(0)
off_row, off_cnt = B * k * 8, 2 * B * k * 8
size = off_cnt + ((B * 4 + 7) // 8) * 8
blob = np.zeros(s * size, np.uint8)
dist = np.sort(rng.random((s, B, k)), axis=2)
dist[:, :, 3] = dist[:, :, 2]  # ties inside a shard
if s > 1: dist[1] = dist[0]     # ... and across shards
rows = rng.integers(0, 1 << 40, (s, B, k))
cnt = rng.integers(0, k + 1, (s, B)).astype(np.int32)
for sh in range(s):
    base = sh * size
    blob[base:base + off_row].view(np.float64)[:] = dist[sh].reshape(-1)
    blob[base + off_row:base + off_cnt].view(np.int64)[:] = rows[sh].reshape(-1)
    blob[base + off_cnt:base + off_cnt + 4 * B].view(np.int32)[:] = cnt[sh]
g = torch.from_numpy(blob).cuda()
o_d = torch.zeros((B, k), dtype=torch.float64, device="cuda"); o_r = torch.zeros((B, k), dtype=torch.int64, device="cuda")
o_c = torch.zeros(B, dtype=torch.int32, device="cuda")
st = torch.cuda.current_stream().cuda_stream
def run():
    nat.check(nat.lib.mir_topk_merge_device(g.data_ptr(), g.data_ptr() + off_row, g.data_ptr() + off_cnt, s, size, B, k, 0,
                                            o_d.data_ptr(), o_r.data_ptr(), o_c.data_ptr(), 0, st))
run(); torch.cuda.synchronize()
# numpy reference: (distance, row) ascending over the valid candidates
for q in range(B):
    cand = sorted((dist[sh, q, p], rows[sh, q, p]) for sh in range(s) for p in range(cnt[sh, q]))[:k]
    assert o_c[q].item() == len(cand)
    assert [(float(o_d[q, i]), int(o_r[q, i])) for i in range(len(cand))] == [(float(a), int(b)) for a, b in cand], q
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
for _ in range(5): run()
e0.record()
for _ in range(200): run()
e1.record(); torch.cuda.synchronize()
print(f"s={s} B={B} k={k}: merge correct; {e0.elapsed_time(e1) / 200 * 1e3:.1f} us per launch (back to back)")
