import sys, time, torch
sys.path.insert(0, "/root/repo")
import bench
from aidial_rag_amd.retrievers.embeddings_index import DeviceIndex
dev = torch.device("cuda:0")
rows = bench.gen_rows(torch, dev, 0, 10_000_000, 384)
torch.cuda.synchronize()
for i in range(2):
    t0 = time.perf_counter()
    ix = DeviceIndex.from_device_ptr(rows.data_ptr(), 10_000_000, 384, 0, stream=torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(f"index build from rows in HBM: {dt*1e3:.1f} ms, int8={ix.scan_stats()['int8_first_stage']}")
    ix.close()
