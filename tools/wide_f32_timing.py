"""float32 d = 1024 shard (the MultimodalRetriever shape as the retriever surface builds it): ms per 64-query pass."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from aidial_rag_amd.retrievers.embeddings_index import DeviceIndex
n, d = int(sys.argv[1]) if len(sys.argv) > 1 else 2_000_000, int(sys.argv[2]) if len(sys.argv) > 2 else 1024
g = torch.Generator(device="cuda"); g.manual_seed(1)
rows = torch.randn((n, d), generator=g, dtype=torch.float32, device="cuda")
ix = DeviceIndex.from_device_ptr(rows.data_ptr(), n, d, 0, stream=torch.cuda.current_stream().cuda_stream)
torch.cuda.synchronize(); del rows
q = torch.randn((64, d), generator=g, dtype=torch.float32, device="cuda").double().contiguous()
o = [torch.zeros((64, 10), dtype=t, device="cuda") for t in (torch.int64, torch.float64)] + [torch.zeros(64, dtype=torch.int32, device="cuda") for _ in range(2)]
ix.profile(True)
for rep in range(25):
    if rep == 5: torch.cuda.synchronize(); ix.profile_read(reset=True); t0 = time.perf_counter()
    ix.search_device(q.data_ptr(), 64, 10, "sqeuclidean_dist", o[0].data_ptr(), o[1].data_ptr(), o[2].data_ptr(), o[3].data_ptr(), stream=torch.cuda.current_stream().cuda_stream)
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 20
ln, ms = ix.profile_read()
print(f"{n} x {d} f32: {dt*1e3:.3f} ms per 64-query step, scan {ms/ln:.3f} ms = {n*d*4/(ms/ln*1e-3)/1e12:.2f} TB/s ({n*d*4/(ms/ln*1e-3)/8e12*100:.1f} % of 8 TB/s), flags {int(o[3].sum())}")
