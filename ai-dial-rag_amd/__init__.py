"""MI355X-native retrieval hot path behind ai-dial-rag's retriever surface.

Everything numeric runs in ``libmiretr.so`` (hand-written HIP for gfx950, C ABI
in ``include/miretr.h``); this package is the thin host-side mirror of the
reference's Python interface for that path.  There is no CPU fallback: the
modules raise at first use when the library or a GPU is missing.
"""

__version__ = "0.1.0"
