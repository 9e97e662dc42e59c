#!/bin/bash
# one rocprofv3 --pmc pass per ARGUMENT (a quoted group of counters that fit one pass) over tools/sieve_stats.py $ARGS
# (default "10000000 256"); prints every counter's mean over the sieve's full-size (second) filter launches
R=${GRAFT_REPO_ROOT:-$PWD}
ARGS=${ARGS:-"10000000 256"}
cd /tmp && export TMPDIR=/tmp
g=0
for grp in "$@"; do
  g=$((g+1))
  rm -rf /tmp/sv_pmcg_$g
  timeout -k 5 200 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d /tmp/sv_pmcg_$g -o p -- python3 $R/tools/sieve_stats.py $ARGS > /tmp/sv_pmcg_$g.log 2>&1 || { echo "$grp: failed"; tail -3 /tmp/sv_pmcg_$g.log; continue; }
  python3 - "$g" <<'PY'
import csv, sys, glob, collections
g = sys.argv[1]
per = collections.defaultdict(lambda: collections.defaultdict(float))
for f in glob.glob(f"/tmp/sv_pmcg_{g}/**/p_counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        n = r["Kernel_Name"]
        if ("sieve_q16_kernel" in n or "sieve_i8_kernel" in n) and "true" not in n.split("<")[1].split(">")[0].split(",")[2]:
            per[r["Counter_Name"]][int(r["Dispatch_Id"])] += float(r["Counter_Value"])
for c, d in sorted(per.items()):
    v = sorted(d.values(), reverse=True)
    v = v[: max(1, len(v) // 2 - 2)]  # the second (full-size) launches
    print(f"{c:28s} {sum(v)/len(v):.5g}", flush=True)
PY
done
