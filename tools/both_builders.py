"""bench.py's `both_builders` leg alone (index build through BM25Retriever.build_index and build_embeddings, each alone and
together as documents.py:188-198 runs them): python tools/both_builders.py [chunks]"""
import asyncio, json, os, sys, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import bench

class A: pass
args = A(); args.encode_chunks = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
with tempfile.TemporaryDirectory() as td:
    r = bench.build_embeddings_leg(np, torch, args, 0, td)
print(json.dumps(r["both_builders"]))
