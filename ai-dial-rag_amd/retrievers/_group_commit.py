"""Request micro-batching at the retriever boundary: host logic, no GPU code."""

import threading


class _GroupCommit:
    """Coalesces concurrent calls (``EmbeddingsIndex.find``, ``BM25Retriever``, ``BgeEncoder.embed_query`` /
    ``embed_documents_numpy``) into shared passes.

    The reference issues one-vector ``find`` calls from many executor threads
    (semantic_retriever.py:54-56, cpu_pools.py:31-34); a pass over the index costs the same for 1 or 96
    queries, so what turns kernel throughput into service QPS is batching at this boundary
    (SURVEY 8(f), rank 4).  No timer and no dedicated thread: the first caller to find no pass in flight
    becomes the leader and runs a pass for everything queued; callers that arrive while a pass is running
    queue up and ride the next one.  A lone caller pays nothing extra.

    Liveness and isolation:
    * a leader serves passes only until ITS OWN result is in, then hands leadership to a waiting rider
      (under sustained load no caller's latency grows with the traffic behind it);
    * ``validate`` runs in the submitting thread before the item is queued, so a malformed item fails its own
      caller and never the other riders of a pass;
    * if a pass dies with a BaseException (KeyboardInterrupt in the leader), the riders of THAT pass are failed
      and leadership is released: nobody waits forever, the queue behind elects a new leader.
    """

    def __init__(self, run_batch, max_batch: int = 96, validate=None):
        self._run = run_batch          # list of b items -> tuple of sequences with leading dimension b
        self._max = max_batch
        self._validate = validate      # item -> item (may raise); called outside the lock, in the caller
        self._cv = threading.Condition()
        self._queue: list = []         # (item, holder); holder = [done, result row or exception]
        self._leader = False
        self.passes = 0
        self.calls = 0

    def submit(self, item):
        if self._validate is not None:
            item = self._validate(item)
        holder = [False, None]
        with self._cv:
            self.calls += 1
            self._queue.append((item, holder))
            while True:
                if holder[0]:
                    return self._unwrap(holder)
                if not self._leader:
                    self._leader = True
                    break
                self._cv.wait()
        batch = []
        try:
            while True:
                with self._cv:
                    if holder[0]:  # own result is in: the next waiting rider takes over
                        break
                    batch, self._queue = self._queue[: self._max], self._queue[self._max :]
                    self.passes += 1
                try:
                    out = self._run([q for q, _ in batch])
                    rows = [tuple(a[i] for a in out) for i in range(len(batch))]
                except Exception as e:  # every rider of this pass sees the failure
                    rows = [e] * len(batch)
                with self._cv:
                    for (_, h), r in zip(batch, rows):
                        h[1] = r
                        h[0] = True
                    batch = []
                    self._cv.notify_all()
        except BaseException:  # e.g. KeyboardInterrupt inside the pass: its riders must not wait forever
            with self._cv:
                for _, h in batch:
                    if not h[0]:
                        h[1], h[0] = RuntimeError("search pass aborted"), True
            raise
        finally:
            with self._cv:
                self._leader = False
                self._cv.notify_all()
        return self._unwrap(holder)

    @staticmethod
    def _unwrap(holder):
        if isinstance(holder[1], Exception):
            raise holder[1]
        return holder[1]
