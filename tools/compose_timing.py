"""Index over a new combination of known documents: flattened on the host and uploaded (the only way before
mir_rows) vs composed device-to-device from per-document blocks already in HBM."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from aidial_rag_amd.retrievers.embeddings_index import DeviceIndex, DeviceRows

n_docs = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
rows = int(sys.argv[2]) if len(sys.argv) > 2 else 2500
rng = np.random.default_rng(0)
docs = [rng.standard_normal((rows, 384), dtype=np.float32) for _ in range(n_docs)]
ids = [np.arange(rows, dtype=np.int64) for _ in range(n_docs)]
gb = n_docs * rows * 384 * 4 / 1e9
t0 = time.perf_counter(); blocks = [DeviceRows.from_host(d, i) for d, i in zip(docs, ids)]; t_up = time.perf_counter() - t0
print(f"{n_docs} documents x {rows} rows = {n_docs*rows/1e6:.2f}M rows, {gb:.2f} GB; first upload of the blocks {t_up*1e3:.0f} ms", flush=True)
for trial in range(2):
    t0 = time.perf_counter()
    flat = DeviceIndex.from_host(np.concatenate(docs), np.concatenate(ids), np.concatenate([np.full(rows, i, np.int32) for i in range(n_docs)]))
    t_flat = time.perf_counter() - t0
    order = rng.permutation(n_docs)[: n_docs - 1].tolist()  # a combination not seen before
    t0 = time.perf_counter()
    comp = DeviceIndex.from_rows([blocks[i] for i in order], list(range(len(order))))
    t_comp = time.perf_counter() - t0
    print(f"host flatten + upload + build {t_flat*1e3:.0f} ms | compose in HBM + build {t_comp*1e3:.1f} ms  ({t_flat/t_comp:.0f}x)", flush=True)
    del flat, comp
