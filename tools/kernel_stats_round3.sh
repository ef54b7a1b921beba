# rocprofv3 --kernel-trace --stats of the default bench command and of --config two-level; the kernel_stats csv files go to gpurun_out/
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd /tmp && export TMPDIR=/tmp
for CFG in "bench --no-two-level" "two_level --config two-level"; do
  set -- $CFG; TAG=$1; shift
  OUT=$R/gpurun_out/stats_$TAG
  rm -rf "$OUT"; mkdir -p "$OUT"
  timeout -s KILL 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT" -- python3 "$R/bench.py" --steps 5 --no-cpu "$@" > "$OUT/line.json" 2> "$OUT/err.log"
  rc=$?; echo "$TAG rc=$rc"; [ $rc -eq 0 ] || { tail -5 "$OUT/err.log"; exit $rc; }
  F=$(find "$OUT" -name "*kernel_stats.csv" | head -1)
  cp "$F" "$R/gpurun_out/r3_p_${TAG}_kernel_stats.csv"
  head -8 "$F"
  python3 -c "import json,sys; j=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print(j['value'], j['ms_per_step'], j['roofline']['kernel_ms'])" "$OUT/line.json"
done
