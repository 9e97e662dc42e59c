"""CPU oracle for the ai-dial-rag retrieval hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``oracle/`` is product code.  Only
``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of
``bench.py`` may import it, and only as the checker / reported baseline.
The product path (``aidial_rag_amd`` -> ``libmiretr.so``) never imports it and
fails loudly when the HIP library is missing.

Each function is a plain numpy / pure-Python restatement of the reference's
algorithm and cites the reference ``file:line`` it follows (paths relative to
the upstream repo root, epam/ai-dial-rag @ 2025-09-05).

Pinning status ("Oracle"):

* vector metrics  - PINNED: checked against every known-answer case of the
  reference's ``tests/test_embeddings_metrics.py`` and, in the build container,
  against the imported reference module itself
  (``tests/golden/make_golden.py`` -> ``tests/golden/metrics_*.npz``).
* index / top-k   - PINNED by the cases of ``tests/test_embeddings_index.py``
  (tie-break, limits, empties) restated in ``tests/golden/index_cases.json``.
* BM25            - parity unpinned upstream: arithmetic lives in third-party
  ``rank-bm25==0.2.2`` which is absent; restated from its published algorithm.
* RRF fusion      - parity unpinned upstream: langchain 0.3.21
  ``EnsembleRetriever.weighted_reciprocal_rank`` restated.
* encoder         - ``transformers.BertModel`` fp32 (third-party, present) is
  the arithmetic oracle; real bge-small-en weights are unavailable offline.
"""
