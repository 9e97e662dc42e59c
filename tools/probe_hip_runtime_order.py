import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sys, os
order = sys.argv[1]
def maps():
    return sorted({l.split()[-1] for l in open('/proc/self/maps') if 'amdhip64' in l or 'hsa-runtime' in l})
if order == "torch_first":
    import torch
    print("torch avail:", torch.cuda.is_available(), torch.cuda.device_count())
    x = torch.ones(4, device="cuda"); print(x.sum().item())
    print(maps())
    from aidial_rag_amd import _native
    print("mir devices:", _native.device_count())
    print(maps())
else:
    from aidial_rag_amd import _native
    print("mir devices:", _native.device_count())
    print(maps())
    import torch
    print(maps())
    print("torch avail:", torch.cuda.is_available(), torch.cuda.device_count())
import numpy as np
from aidial_rag_amd.retrievers.embeddings_index import DeviceIndex
dev = DeviceIndex.from_host(np.random.default_rng(0).standard_normal((1000,384)).astype(np.float32))
print(dev.search(np.ones((1,384)), 3, "inner_product")[2])
import torch
x = torch.ones(4, device="cuda"); print("torch ok", x.sum().item())
