#!/bin/bash
# PMC passes over the diagnostic FFN harness (one counter per pass): bash tools/run_ffn_pmc.sh COUNTER...
# build flags in $ABL as for run_ffn_stamps.sh
set -e
R=${GRAFT_REPO_ROOT:-$PWD}
/opt/rocm/bin/hipcc -O3 -std=c++17 -ffp-contract=on -mllvm -amdgpu-mfma-vgpr-form=1 --offload-arch=gfx950 -DFFN_RING=${RING:-4} $ABL -I $R/ai-dial-rag_amd/csrc -I $R/include $R/tools/ffn_stamps.hip -o /tmp/ffn_stamps
cd /tmp && export TMPDIR=/tmp
for c in "$@"; do
  rm -rf /tmp/ffn_pmc_$c
  timeout -k 5 120 rocprofv3 --pmc $c --kernel-trace --output-format csv -d /tmp/ffn_pmc_$c -o p -- /tmp/ffn_stamps > /tmp/ffn_pmc_$c.log 2>&1 || { echo "$c: failed"; tail -3 /tmp/ffn_pmc_$c.log; continue; }
  python3 - "$c" <<'PY'
import csv, sys, glob
c = sys.argv[1]
v = [float(r["Counter_Value"]) for f in glob.glob(f"/tmp/ffn_pmc_{c}/**/p_counter_collection.csv", recursive=True) for r in csv.DictReader(open(f)) if "ffn_ln_kernel" in r["Kernel_Name"] and r["Counter_Name"] == c]
print(f"{c:32s} launches={len(v)} mean={sum(v)/max(len(v),1):.4g}")
PY
done
