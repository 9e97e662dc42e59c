#!/bin/bash
# one rocprofv3 --pmc pass per ARGUMENT (a quoted group of counters that fit one pass, e.g. "SQ_WAVE_CYCLES SQ_WAIT_ANY") over
# tools/encoder_throughput.py 2048; prints per encoder throughput kernel the mean per launch of every counter of the group
R=${GRAFT_REPO_ROOT:-$PWD}
cd /tmp && export TMPDIR=/tmp
g=0
for grp in "$@"; do
  g=$((g+1))
  rm -rf /tmp/enc_pmcg_$g
  timeout -k 5 200 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d /tmp/enc_pmcg_$g -o p -- python3 $R/tools/encoder_throughput.py 2048 > /tmp/enc_pmcg_$g.log 2>&1 || { echo "$grp: failed"; tail -3 /tmp/enc_pmcg_$g.log; continue; }
  python3 - "$g" <<'PY'
import csv, sys, glob, collections
g = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(f"/tmp/enc_pmcg_{g}/**/p_counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "enc::" in r["Kernel_Name"] and "small" not in r["Kernel_Name"]:
            agg[r["Counter_Name"]][r["Kernel_Name"].split("(")[0].replace("void ", "").replace("mir::enc::", "")[:24]].append(float(r["Counter_Value"]))
for c, per in sorted(agg.items()):
    print(f"{c:28s} " + "  ".join(f"{k}={sum(v)/len(v):.4g}" for k, v in sorted(per.items()) if len(v) > 50), flush=True)
PY
done
