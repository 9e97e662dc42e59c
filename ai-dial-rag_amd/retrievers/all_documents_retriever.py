"""The "everything fits in the prompt" shortcut of create_retriever
(aidial_rag/retrieval_chain.py:201,246-250; aidial_rag/retrievers/all_documents_retriever.py:10-66).

When the chunks of all documents, with the per-chunk attribute header the prompt adds, stay within 12 000
characters, the reference does not search at all: the retriever returns every chunk in (document, chunk) order.
Host logic only - there is nothing to compute; it is part of the path because it decides whether the search
kernels run for a request.
"""

from typing import List, Optional

from ..index_record import Document, RetrievalType, to_metadata_doc


def format_attributes(id: int, page_number: Optional[int], source_url: Optional[str]) -> str:
    """The attribute header of one chunk in the prompt (qa_chain.py:52-63 upstream); only its LENGTH matters here."""
    attributes = [("id", str(id))]
    if page_number is not None:
        attributes.append(("page_number", str(page_number)))
    if source_url:
        attributes.append(("source", source_url))
    return " ".join(f"{k}='{v}'" for k, v in attributes)


class AllDocumentsRetriever:
    _MAX_LENGTH_IN_BYTES = 12000
    _CHUNK_PROMPT_OVERHEAD = 30

    def __init__(self, metadata_chunks: List[Document]):
        self.metadata_chunks = metadata_chunks

    @staticmethod
    def _estimated_size(i: int, chunk) -> int:
        return (len(chunk.text)
                + len(format_attributes(id=i, page_number=chunk.metadata.get("page_number"), source_url=chunk.metadata.get("source")))
                + AllDocumentsRetriever._CHUNK_PROMPT_OVERHEAD)

    @staticmethod
    def is_within_limit(document_records) -> bool:
        total = sum(AllDocumentsRetriever._estimated_size(i, chunk)
                    for i, chunk in enumerate(chunk for doc in document_records for chunk in doc.chunks))
        return total <= AllDocumentsRetriever._MAX_LENGTH_IN_BYTES

    @classmethod
    def from_doc_records(cls, document_records=None) -> "AllDocumentsRetriever":
        document_records = document_records or []
        return cls([to_metadata_doc(i, j, RetrievalType.TEXT) for i, doc in enumerate(document_records) for j in range(len(doc.chunks))])

    def _get_relevant_documents(self, query: str, *args, **kwargs) -> List[Document]:
        return self.metadata_chunks

    async def _aget_relevant_documents(self, query: str, *args, **kwargs) -> List[Document]:
        return self.metadata_chunks

    invoke = _get_relevant_documents
