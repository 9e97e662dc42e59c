"""Semantic retriever, same surface as aidial_rag/retrievers/semantic_retriever.py:23-66:
encode the query with the BGE encoder, then ``EmbeddingsIndex.find`` (default metric
sqeuclidean, embeddings_index.py:43).  Both halves run on the GPU."""

import asyncio
from typing import List, Optional

import numpy as np

from ..embeddings import embeddings as emb
from ..index_record import Document, RetrievalType
from .embeddings_index import EmbeddingsIndex, create_index_by_chunk, pack_simple_embeddings


class SemanticRetriever:
    def __init__(self, index: EmbeddingsIndex, encoder: Optional[emb.BgeEncoder] = None):
        self.index = index
        self._encoder = encoder

    def _enc(self) -> emb.BgeEncoder:
        return self._encoder if self._encoder is not None else emb.bge_embedding_impl()

    @classmethod
    def from_doc_records(cls, document_records, k: int = 1, encoder: Optional[emb.BgeEncoder] = None, device: int = 0) -> "SemanticRetriever":
        # semantic_retriever.py:26-41
        sources = [doc.embeddings_index for doc in document_records if doc.embeddings_index]
        # same source objects as an earlier request -> the index already in HBM (retrievers/_device_cache.py);
        # the per-chunk flattening loop runs only on a miss
        return cls(index=EmbeddingsIndex(retrieval_type=RetrievalType.TEXT,
                                         indexes=lambda: [create_index_by_chunk(src) for src in sources], limit=k,
                                         device=device, cache_sources=sources), encoder=encoder)

    def _find_relevant_documents(self, query_emb: np.ndarray) -> List[Document]:
        return self.index.find(query=query_emb)

    def _get_relevant_documents(self, query: str, *args, **kwargs) -> List[Document]:
        # semantic_retriever.py:46-50: np.array(List[float]) -> the float64 query of the live path
        return self._find_relevant_documents(np.array(self._enc().embed_query(query)))

    async def _aget_relevant_documents(self, query: str, *args, **kwargs) -> List[Document]:
        loop = asyncio.get_running_loop()
        query_emb = np.array(await loop.run_in_executor(None, self._enc().embed_query, query))
        return await loop.run_in_executor(None, self._find_relevant_documents, query_emb)

    def get_relevant_documents_batch(self, queries: List[str]) -> List[List[Document]]:
        """B queries: one encoder pass, one index pass."""
        enc = self._enc()
        texts = [emb.BGE_QUERY_INSTRUCTION_EN + q.replace("\n", " ") for q in queries]
        q = enc.encode_ids(enc._tokenize(texts)).astype(np.float64)
        return self.index.find_batch(q)

    invoke = _get_relevant_documents

    @staticmethod
    async def build_index(chunks, stageio=None, encoder: Optional[emb.BgeEncoder] = None):
        # semantic_retriever.py:58-66
        if encoder is not None:
            vecs = await asyncio.get_running_loop().run_in_executor(None, encoder.embed_documents_numpy, [c.text for c in chunks])
        else:
            vecs = await emb.build_embeddings((c.text for c in chunks), stageio)
        return pack_simple_embeddings(vecs)
