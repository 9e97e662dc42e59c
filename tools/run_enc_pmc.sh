#!/bin/bash
# one rocprofv3 --pmc pass per counter over tools/encoder_throughput.py; prints per encoder kernel the mean per launch
R=${GRAFT_REPO_ROOT:-$PWD}
cd /tmp && export TMPDIR=/tmp
for c in "$@"; do
  rm -rf /tmp/enc_pmc_$c
  timeout -k 5 150 rocprofv3 --pmc $c --kernel-trace --output-format csv -d /tmp/enc_pmc_$c -o p -- python3 $R/tools/encoder_throughput.py 2048 > /tmp/enc_pmc_$c.log 2>&1 || { echo "$c: failed"; tail -3 /tmp/enc_pmc_$c.log; continue; }
  python3 - "$c" <<'PY'
import csv, sys, glob, collections
c = sys.argv[1]
agg = collections.defaultdict(list)
for f in glob.glob(f"/tmp/enc_pmc_{c}/**/p_counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "enc::" in r["Kernel_Name"] and r["Counter_Name"] == c and "small" not in r["Kernel_Name"]:
            agg[r["Kernel_Name"].split("(")[0].replace("void ", "").replace("mir::enc::", "")[:24]].append(float(r["Counter_Value"]))
print(f"{c:28s} " + "  ".join(f"{k}={sum(v)/len(v):.4g}" for k, v in sorted(agg.items()) if len(v) > 50))
PY
done
