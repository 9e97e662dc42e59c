import os, sys, time, asyncio, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bench
from aidial_rag_amd.embeddings import embeddings as emb
from aidial_rag_amd.embeddings.wordpiece import WordPieceTokenizer
if len(sys.argv) > 1: emb.BUILD_IN_FLIGHT = int(sys.argv[1])   # experiment: outer batches in flight
if len(sys.argv) > 2: sys.setswitchinterval(float(sys.argv[2]))  # experiment: the GIL's switch interval (default 0.005 s)
rng = np.random.default_rng(7)
letters = "abcdefghijklmnopqrstuvwxyz"
words = ["".join(rng.choice(list(letters), rng.integers(2, 9))) for _ in range(20000)]
vocab = ["[PAD]"] + [f"[unused{i}]" for i in range(99)] + ["[UNK]", "[CLS]", "[SEP]", "[MASK]"] + list(letters) + ["##" + c for c in letters] + sorted(set(words))
td = tempfile.mkdtemp()
open(os.path.join(td, "vocab.txt"), "w").write("\n".join(vocab) + "\n")
tok = WordPieceTokenizer.from_vocab_file(os.path.join(td, "vocab.txt"))
enc = emb.BgeEncoder.from_state_dict(bench.random_bge_small_state_dict(np), tokenizer=tok, device=0)
emb.set_bge_embedding_impl(enc)
n = 8192
lens = np.clip(np.round(rng.normal(218, 60, n)), 6, 510).astype(np.int64)
texts = [" ".join(rng.choice(words, L)) for L in lens]
asyncio.run(emb.build_embeddings(texts[:256]))
for rep in range(3):
    t0 = time.perf_counter(); out = asyncio.run(emb.build_embeddings(texts)); dt = time.perf_counter() - t0
    print(f"build_embeddings: {n/dt:.0f} chunks/s ({dt*1e3:.0f} ms)")
for rep in range(3):
    t0 = time.perf_counter(); ids = enc._tokenize(texts); t1 = time.perf_counter(); o = enc.encode_ids(ids); t2 = time.perf_counter()
    print(f"tokenize all {1e3*(t1-t0):.0f} ms, encode_ids {1e3*(t2-t1):.0f} ms -> {n/(t2-t0):.0f} chunks/s")
t0 = time.perf_counter(); o = enc.embed_documents_numpy(texts); dt = time.perf_counter() - t0
print(f"embed_documents_numpy (one call): {n/dt:.0f} chunks/s")
