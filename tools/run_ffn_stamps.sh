#!/bin/bash
# builds and runs the diagnostic FFN harness on the GPU box
set -e
for ring in ${RINGS:-8}; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 -ffp-contract=on -fno-slp-vectorize -mllvm -amdgpu-mfma-vgpr-form=1 --offload-arch=gfx950 -DFFN_RING=$ring $ABL -I ai-dial-rag_amd/csrc -Rpass-analysis=kernel-resource-usage tools/ffn_stamps.hip -o /tmp/ffn_stamps 2>&1 | grep -E "VGPRs Spill|ScratchSize" | head -2
  echo "== FFN_RING=$ring"
  timeout -k 5 120 /tmp/ffn_stamps "$@"
done
