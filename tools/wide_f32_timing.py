"""float32 d = 1024 shard (the MultimodalRetriever shape as the retriever surface builds it): ms per B-query step (argv[3],
default 64).  MIR_NO_SIEVE_WIDE=1 (read at index build) selects the K-split list scan over the hi/lo image instead of the
sieve's bf16 filter over the hi-only image (round 4): run both for the A/B."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from aidial_rag_amd.retrievers.embeddings_index import DeviceIndex
n, d = int(sys.argv[1]) if len(sys.argv) > 1 else 2_000_000, int(sys.argv[2]) if len(sys.argv) > 2 else 1024
g = torch.Generator(device="cuda"); g.manual_seed(1)
rows = torch.randn((n, d), generator=g, dtype=torch.float32, device="cuda")
ix = DeviceIndex.from_device_ptr(rows.data_ptr(), n, d, 0, stream=torch.cuda.current_stream().cuda_stream)
torch.cuda.synchronize(); del rows
B = int(sys.argv[3]) if len(sys.argv) > 3 else 64
q = torch.randn((B, d), generator=g, dtype=torch.float32, device="cuda").double().contiguous()
o = [torch.zeros((B, 10), dtype=t, device="cuda") for t in (torch.int64, torch.float64)] + [torch.zeros(B, dtype=torch.int32, device="cuda") for _ in range(2)]
ix.profile(True)
for rep in range(25):
    if rep == 5: torch.cuda.synchronize(); ix.profile_read(reset=True); t0 = time.perf_counter()
    ix.search_device(q.data_ptr(), B, 10, "sqeuclidean_dist", o[0].data_ptr(), o[1].data_ptr(), o[2].data_ptr(), o[3].data_ptr(), stream=torch.cuda.current_stream().cuda_stream)
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 20
ln, ms = ix.profile_read()
print(f"{n} x {d} f32, {'K-split list scan' if os.environ.get('MIR_NO_SIEVE_WIDE') else 'wide sieve'}: {dt*1e3:.3f} ms per {B}-query step = {B/dt:.0f} QPS, "
      f"bracket {ms/ln:.3f} ms = {n*d*4/(ms/ln*1e-3)/1e12:.2f} TB/s of float32 rows ({n*d*4/(ms/ln*1e-3)/8e12*100:.1f} % of 8 TB/s), index {ix.hbm_bytes()/1e9:.2f} GB, flags {int(o[3].sum())}, {ix.scan_stats()}")
