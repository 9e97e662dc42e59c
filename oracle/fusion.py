"""Oracle: weighted reciprocal-rank fusion of retriever result lists.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

PARITY UNPINNED upstream at list level.  The arithmetic is third-party:
langchain 0.3.21 ``EnsembleRetriever.weighted_reciprocal_rank``
(poetry.lock:2090-2091), called from aidial_rag/retrieval_chain.py:239-245
with all weights 1.0 and the default c = 60.  The key of an item is its
``page_content`` = ``"{doc_id}_{chunk_id}"`` (aidial_rag/index_record.py:33-34),
i.e. the (doc_id, chunk_id) pair.

Algorithm restated: score[key] += weight / (rank + c) with rank from 1 over
every list (an item occurring twice in one list is credited twice); items are
de-duplicated in first-seen order over the chained lists; a stable sort by
score, descending, keeps first-seen order among equal scores.
"""

from typing import Hashable, List, Sequence


def weighted_reciprocal_rank(
    lists: Sequence[Sequence[Hashable]], weights: Sequence[float], c: int = 60
) -> List[Hashable]:
    if len(lists) != len(weights):
        raise ValueError("Number of rank lists must be equal to the number of weights.")
    score = {}
    for items, w in zip(lists, weights):
        for rank, key in enumerate(items, start=1):
            score[key] = score.get(key, 0.0) + w / (rank + c)
    seen = set()
    unique = []
    for items in lists:
        for key in items:
            if key not in seen:
                seen.add(key)
                unique.append(key)
    return sorted(unique, key=lambda k: score[k], reverse=True)


def rrf_scores(lists, weights, c: int = 60):
    """The score table alone (float64), for numeric comparison."""
    score = {}
    for items, w in zip(lists, weights):
        for rank, key in enumerate(items, start=1):
            score[key] = score.get(key, 0.0) + w / (rank + c)
    return score
