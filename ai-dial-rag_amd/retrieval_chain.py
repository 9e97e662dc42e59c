"""Retriever assembly, the hot-path part of aidial_rag/retrieval_chain.py:193-252.

``create_retriever(document_records, ...)`` builds the member retrievers with
k = 7 each and fuses them with weights 1.0 (retrieval_chain.py:203-245); when
all chunks fit the prompt budget it returns the no-search
``AllDocumentsRetriever`` instead (:201,246-250).
"""

from typing import Callable, List, Optional

from .retrievers.all_documents_retriever import AllDocumentsRetriever
from .retrievers.bm25_retriever import BM25Retriever
from .retrievers.embeddings_metrics import Metric
from .retrievers.ensemble_retriever import EnsembleRetriever
from .retrievers.page_retrievers import DescriptionRetriever, MultimodalRetriever
from .retrievers.semantic_retriever import SemanticRetriever

RETRIEVER_K = 7  # retrieval_chain.py:203,211,224,233


def create_retriever(document_records, encoder=None, keywords_preprocess: Optional[Callable[[str], List[str]]] = None,
                     multimodal_embed_query: Optional[Callable[[str], List[float]]] = None,
                     multimodal_metric: Metric = Metric.SQEUCLIDEAN_DIST, device: int = 0):
    if AllDocumentsRetriever.is_within_limit(document_records):
        return AllDocumentsRetriever.from_doc_records(document_records)
    retrievers = [SemanticRetriever.from_doc_records(document_records, RETRIEVER_K, encoder=encoder, device=device)]
    weights = [1.0]
    if BM25Retriever.has_index(document_records):
        retrievers.append(BM25Retriever.from_doc_records(document_records, RETRIEVER_K, device=device, preprocess=keywords_preprocess))
        weights.append(1.0)
    if MultimodalRetriever.has_index(document_records):
        retrievers.append(MultimodalRetriever.from_doc_records(document_records, RETRIEVER_K, metric=multimodal_metric,
                                                               embed_query=multimodal_embed_query, device=device))
        weights.append(1.0)
    if DescriptionRetriever.has_index(document_records):
        retrievers.append(DescriptionRetriever.from_doc_records(document_records, RETRIEVER_K, encoder=encoder, device=device))
        weights.append(1.0)
    return EnsembleRetriever(retrievers=retrievers, weights=weights)
