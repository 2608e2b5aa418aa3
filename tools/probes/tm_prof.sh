cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/tmp1 -- python3 $R/tools/bound_run.py --steps 400 --param fuse_tm=1 --param overlap_halves=0 --param tm_rows=64 > /tmp/tmp1.log 2>&1
f=$(find /tmp/tmp1 -name "*kernel_stats.csv" | head -1)
python3 - "$f" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if float(r["TotalDurationNs"]) > 1e6: print("%-60s calls %5s avg %8.2f us min %8.2f max %8.2f" % (r["Name"].split("(")[0][:60], r["Calls"], float(r["AverageNs"])/1e3, float(r["MinNs"])/1e3, float(r["MaxNs"])/1e3))
PY
