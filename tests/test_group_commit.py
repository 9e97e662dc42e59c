"""Host logic of the request micro-batching at the find() boundary: no GPU needed,
the search pass is a stand-in that sleeps like one."""

import threading
import time

import numpy as np

from aidial_rag_amd.retrievers._group_commit import _GroupCommit


def _fake_pass(log):
    def run(items):
        q = np.stack(items)
        log.append(len(q))
        time.sleep(0.01)  # a pass costs the same for 1 or 96 queries
        return (q[:, 0] * 2.0, np.arange(len(q)))
    return run


def test_lone_caller_runs_its_own_pass():
    log = []
    gc = _GroupCommit(_fake_pass(log))
    out = gc.submit(np.array([3.0, 1.0]))
    assert out[0] == 6.0 and log == [1] and gc.passes == 1


def test_concurrent_callers_share_passes_and_get_their_own_rows():
    log = []
    gc = _GroupCommit(_fake_pass(log), max_batch=16)
    results = {}

    def worker(i):
        for j in range(5):
            v = float(100 * i + j)
            results[(i, j)] = gc.submit(np.array([v, 0.0]))[0]

    threads = [threading.Thread(target=worker, args=(i,)) for i in range(24)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert all(results[(i, j)] == 2.0 * (100 * i + j) for i in range(24) for j in range(5))
    assert sum(log) == 120 and gc.calls == 120
    assert gc.passes < 60 and max(log) <= 16  # riders were batched, never beyond max_batch


def test_a_failing_pass_reaches_every_rider_and_the_queue_recovers():
    calls = []

    def run(items):
        calls.append(len(items))
        if len(calls) == 1:
            raise ValueError("boom")
        return (np.stack(items)[:, 0],)

    gc = _GroupCommit(run)
    try:
        gc.submit(np.array([1.0]))
        raise AssertionError("expected ValueError")
    except ValueError:
        pass
    assert gc.submit(np.array([7.0]))[0] == 7.0
