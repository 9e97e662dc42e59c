"""Result currency of the retrievers.

Mirrors aidial_rag/index_record.py:18-38 (RetrievalType, ChunkMetadata,
to_metadata_doc).  The reference returns langchain ``Document`` objects; when
langchain is importable its class is used, otherwise a structurally identical
stand-in (same two fields, same equality) so results compare the same way.
"""

from enum import Enum
from typing import Any, Dict

try:  # pragma: no cover - langchain is absent from the build image
    from langchain.schema import Document  # type: ignore
except Exception:

    class Document:  # minimal langchain.schema.Document: page_content + metadata
        __slots__ = ("page_content", "metadata")

        def __init__(self, page_content: str, metadata: Dict[str, Any] | None = None):
            self.page_content = page_content
            self.metadata = metadata if metadata is not None else {}

        def __eq__(self, other):
            return (
                isinstance(other, Document)
                and self.page_content == other.page_content
                and self.metadata == other.metadata
            )

        def __repr__(self):
            return f"Document(page_content={self.page_content!r}, metadata={self.metadata!r})"


class RetrievalType(str, Enum):  # index_record.py:18-20 (StrEnum upstream)
    TEXT = "text"
    IMAGE = "image"

    def __str__(self) -> str:
        return str(self.value)


def to_metadata_doc(doc_id: int, chunk_id: int, retrieval_type: RetrievalType) -> Document:
    # index_record.py:29-38: EnsembleRetriever keys documents by page_content
    return Document(
        page_content=f"{doc_id}_{chunk_id}",
        metadata={"doc_id": doc_id, "chunk_id": chunk_id, "retrieval_type": retrieval_type},
    )
