#!/bin/bash
# rebuilds vec_index.o with each argument as VEC_EXTRA and runs the command in $CMD (default: the C5 shard timing)
R=${GRAFT_REPO_ROOT:-$PWD}
CMD=${CMD:-"python $R/tools/c5_timing.py 6250000"}
# whatever happens, the tree ends with the DEFAULT build (a later `make` would otherwise keep shipping an experiment)
trap 'make -C $R/ai-dial-rag_amd/csrc -B build/vec_index.o VEC_EXTRA= > /dev/null 2>&1; make -C $R/ai-dial-rag_amd/csrc > /dev/null 2>&1' EXIT
for v in "$@"; do
  make -C $R/ai-dial-rag_amd/csrc -B build/vec_index.o VEC_EXTRA="$v" > /tmp/vec_build.log 2>&1 && make -C $R/ai-dial-rag_amd/csrc >> /tmp/vec_build.log 2>&1 || { grep -E "error" /tmp/vec_build.log | head -5; continue; }
  echo "### $v"
  timeout -k 10 200 $CMD 2>&1 | grep -v "amdgpu.ids"
done
