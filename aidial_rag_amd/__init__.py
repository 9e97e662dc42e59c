"""Importable name for the package that lives in ``ai-dial-rag_amd/``.

The directory name required for this repository contains hyphens and cannot be
an import name, so this stub points the package path there and runs the real
``__init__``.  ``import aidial_rag_amd.retrievers.embeddings_index`` mirrors
``import aidial_rag.retrievers.embeddings_index`` of the reference.
"""

import os as _os

_real = _os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))), "ai-dial-rag_amd")
__path__ = [_real]
with open(_os.path.join(_real, "__init__.py")) as _f:
    exec(compile(_f.read(), _os.path.join(_real, "__init__.py"), "exec"))
