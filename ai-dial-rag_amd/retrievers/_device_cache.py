"""Device-resident indexes that outlive a request: host logic, no GPU code.

The reference rebuilds its numpy matrix and its whole BM25 model inside every request
(``from_doc_records``: semantic_retriever.py:26-41, bm25_retriever.py:64-79) from DocumentRecords that it
keeps in an LRU cache of its own (index_storage.py:28-29,57-66), so consecutive requests over the same
documents hand over the SAME Python objects.  The key here is therefore the identity of those source
objects (``doc.embeddings_index`` / ``doc.text_index``), which the entry keeps alive so that an id cannot be
recycled while it is cached.  Least-recently-used entries are dropped once the cached indexes exceed a
budget of HBM bytes; a dropped index is freed when the last retriever using it is gone.

Kinds: "vector" (a composed index over a tuple of documents), "rows" (ONE document's rows in HBM, the
blocks a vector index over any new combination of documents is composed from device-to-device:
embeddings_index.py `_upload`), "bm25" (a BM25 model over a tuple of documents).
"""

import os
import threading
from collections import OrderedDict
from typing import Callable, Hashable, Sequence, Tuple

_BUDGET_BYTES = int(float(os.environ.get("AIDIAL_RAG_AMD_INDEX_CACHE_GB", "64")) * (1 << 30))


class DeviceCache:
    def __init__(self, budget_bytes: int = _BUDGET_BYTES):
        self.budget = budget_bytes
        self._lock = threading.Lock()
        self._entries: "OrderedDict[Hashable, Tuple[object, int, tuple]]" = OrderedDict()  # key -> (value, bytes, sources)
        self.hits = 0
        self.misses = 0
        self.by_kind: dict = {}  # kind -> [hits, misses]

    def get_or_build(self, kind: str, device: int, sources: Sequence[object], build: Callable[[], Tuple[object, int]]):
        """`sources`: the per-document source objects, in order.  `build() -> (value, hbm_bytes)`."""
        key = (kind, device, tuple(id(s) for s in sources))
        with self._lock:
            hit = self._entries.get(key)
            per = self.by_kind.setdefault(kind, [0, 0])
            if hit is not None:
                self._entries.move_to_end(key)
                self.hits += 1
                per[0] += 1
                return hit[0]
            self.misses += 1
            per[1] += 1
        value, nbytes = build()  # outside the lock: uploads take a while, other keys must not wait
        with self._lock:
            if key not in self._entries:
                self._entries[key] = (value, int(nbytes), tuple(sources))
                total = sum(e[1] for e in self._entries.values())
                while total > self.budget and len(self._entries) > 1:
                    _, (_, freed, _) = self._entries.popitem(last=False)
                    total -= freed
            return self._entries[key][0]

    def clear(self):
        with self._lock:
            self._entries.clear()

    def __len__(self):
        return len(self._entries)


CACHE = DeviceCache()
