"""Vector paths of the page-image retrievers.

``MultimodalRetriever`` (aidial_rag/retrievers/multimodal_retriever.py:96-153) and
``DescriptionRetriever`` (aidial_rag/retrievers/description_retriever/
description_retriever.py:92-127) are both "query vector -> EmbeddingsIndex by page
-> image Documents".  Only that part is on the hot path:
the remote multimodal embedding model and the vision-LLM page descriptions are
network services and stay with the caller, who supplies the query vector
(multimodal) or uses the local BGE encoder (descriptions).
"""

from typing import Callable, List, Optional

import numpy as np

from ..embeddings import embeddings as emb
from ..index_record import Document, RetrievalType
from .embeddings_index import EmbeddingsIndex, create_index_by_page
from .embeddings_metrics import Metric


class MultimodalRetriever:
    def __init__(self, index: EmbeddingsIndex, embed_query: Optional[Callable[[str], List[float]]] = None):
        self.index = index
        self._embed_query = embed_query  # stands in for MultimodalEmbeddings(dial_config, model).embed_query

    @staticmethod
    def has_index(document_records) -> bool:
        return any(doc.multimodal_embeddings_index is not None for doc in document_records)

    @classmethod
    def from_doc_records(cls, document_records, k: int = 1, metric: Metric = Metric.SQEUCLIDEAN_DIST,
                         embed_query: Optional[Callable[[str], List[float]]] = None, device: int = 0) -> "MultimodalRetriever":
        # multimodal_retriever.py:108-131: page embeddings expanded to the chunks of each page; metric from the index config
        # (shared across requests over the same records: retrievers/_device_cache.py; the chunk list is part of
        # the key because it decides how page rows are repeated)
        docs = list(document_records)
        return cls(EmbeddingsIndex(retrieval_type=RetrievalType.IMAGE,
                                   indexes=lambda: [create_index_by_page(doc.chunks, doc.multimodal_embeddings_index) for doc in docs],
                                   metric=metric, limit=k, device=device,
                                   cache_sources=[x for doc in docs for x in (doc.chunks, doc.multimodal_embeddings_index)]), embed_query)

    def _find_relevant_documents(self, query_emb: np.ndarray) -> List[Document]:
        return self.index.find(query=query_emb)

    def _get_relevant_documents(self, query: str, *args, **kwargs) -> List[Document]:
        if self._embed_query is None:
            raise RuntimeError("MultimodalRetriever needs an embed_query callable (the DIAL multimodal model is a remote service)")
        return self._find_relevant_documents(np.array(self._embed_query(query)))

    invoke = _get_relevant_documents


class DescriptionRetriever:
    def __init__(self, index: EmbeddingsIndex, encoder: Optional[emb.BgeEncoder] = None):
        self.index = index
        self._encoder = encoder

    @staticmethod
    def has_index(document_records) -> bool:
        return any(doc.description_embeddings_index is not None for doc in document_records)

    @classmethod
    def from_doc_records(cls, document_records, k: int = 4, encoder: Optional[emb.BgeEncoder] = None, device: int = 0) -> "DescriptionRetriever":
        # description_retriever.py:95-112
        docs = list(document_records)
        return cls(EmbeddingsIndex(retrieval_type=RetrievalType.IMAGE,
                                   indexes=lambda: [create_index_by_page(doc.chunks, doc.description_embeddings_index) for doc in docs],
                                   limit=k, device=device,
                                   cache_sources=[x for doc in docs for x in (doc.chunks, doc.description_embeddings_index)]), encoder)

    def _find_relevant_documents(self, query_emb: np.ndarray) -> List[Document]:
        return self.index.find(query=query_emb)

    def _get_relevant_documents(self, query: str, *args, **kwargs) -> List[Document]:
        enc = self._encoder if self._encoder is not None else emb.bge_embedding_impl()
        return self._find_relevant_documents(np.array(enc.embed_query(query)))

    invoke = _get_relevant_documents
