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


def test_leader_returns_as_soon_as_its_own_result_is_in():
    """Sustained load: the first caller must not keep serving everyone else's passes."""
    gate = threading.Event()
    served = []

    def run(items):
        served.append(list(items))
        if len(served) == 1:
            gate.set()
        time.sleep(0.02)
        return (np.asarray(items, dtype=np.float64),)

    gc = _GroupCommit(run, max_batch=4)
    stop = threading.Event()

    def flood():
        gate.wait()
        while not stop.is_set():
            gc.submit(1.0)

    floods = [threading.Thread(target=flood) for _ in range(8)]
    for t in floods:
        t.start()
    t0 = time.perf_counter()
    assert gc.submit(5.0)[0] == 5.0
    took = time.perf_counter() - t0
    time.sleep(0.1)  # the flood keeps going without the first leader
    n_after = len(served)
    time.sleep(0.1)
    stop.set()
    for t in floods:
        t.join()
    assert took < 0.1, f"the leader was held for {took:.3f}s"
    assert len(served) > n_after > 1  # leadership was handed over and passes continued


def test_validation_fails_only_the_offending_caller():
    def validate(x):
        if x < 0:
            raise ValueError("negative")
        return float(x)

    log = []

    def run(items):
        log.append(list(items))
        time.sleep(0.01)
        return (np.asarray(items) * 2,)

    gc = _GroupCommit(run, validate=validate)
    results, errors = {}, {}

    def worker(i):
        try:
            results[i] = gc.submit(i - 3)[0]
        except ValueError as e:
            errors[i] = str(e)

    threads = [threading.Thread(target=worker, args=(i,)) for i in range(12)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert sorted(errors) == [0, 1, 2] and all(results[i] == 2.0 * (i - 3) for i in range(3, 12))
    assert all(x >= 0 for b in log for x in b)


def test_base_exception_in_a_pass_fails_its_riders_and_releases_leadership():
    class Stop(BaseException):
        pass

    started = threading.Event()
    release = threading.Event()
    n = [0]

    def run(items):
        n[0] += 1
        if n[0] == 1:
            started.set()
            release.wait()
            raise Stop()
        return (np.asarray(items, dtype=np.float64),)

    gc = _GroupCommit(run)
    out = {}

    def leader():
        try:
            gc.submit(1.0)
        except Stop:
            out["leader"] = "stop"

    def rider():
        try:
            out["rider"] = gc.submit(2.0)[0]
        except RuntimeError as e:
            out["rider"] = str(e)

    tl = threading.Thread(target=leader)
    tl.start()
    started.wait()
    tr = threading.Thread(target=rider)  # queued behind the dying pass: must be served by a NEW leader (itself)
    tr.start()
    time.sleep(0.05)
    release.set()
    tl.join(2)
    tr.join(2)
    assert not tl.is_alive() and not tr.is_alive()
    assert out["leader"] == "stop" and out["rider"] == 2.0
