cd /tmp && export TMPDIR=/tmp
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
for pe in 1000 100 10; do
  rm -rf /tmp/ps_$pe
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ps_$pe -o kt -- python3 $ROOT/bench.py --records 100000000 --steps 3 --warmup 1 --no-cpu-baseline --plant-every $pe > /tmp/ps_$pe.log 2>&1
  echo "== plant_every=$pe"; python3 - <<PY
import csv,glob
for f in glob.glob("/tmp/ps_$pe/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "mk_" in r["Name"]: print("  %-50.50s calls=%s avg_us=%.1f" % (r["Name"], r["Calls"], float(r["AverageNs"])/1e3))
PY
done
