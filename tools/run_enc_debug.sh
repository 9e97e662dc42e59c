#!/bin/bash
# rebuilds the encoder objects with each argument as ENC_EXTRA and runs tools/enc_debug_layers.py (per-sequence parity after one layer)
R=${GRAFT_REPO_ROOT:-$PWD}
trap 'make -C $R/ai-dial-rag_amd/csrc -B build/encoder.o build/encoder_attention.o build/encoder_ffn.o ENC_EXTRA= -j8 > /dev/null 2>&1; make -C $R/ai-dial-rag_amd/csrc > /dev/null 2>&1' EXIT
for v in "$@"; do
  make -C $R/ai-dial-rag_amd/csrc -B build/encoder.o build/encoder_attention.o build/encoder_ffn.o ENC_EXTRA="$v" -j8 > /tmp/enc_build.log 2>&1 && make -C $R/ai-dial-rag_amd/csrc >> /tmp/enc_build.log 2>&1 || { tail -5 /tmp/enc_build.log; exit 1; }
  echo "### $v"
  timeout -k 5 200 python3 $R/tools/enc_debug_layers.py 1 2>&1 | tail -2
done
