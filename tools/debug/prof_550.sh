cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/pf_550 -- python3 $GRAFT_REPO_ROOT/tools/msda_fused_bench.py --kinds 550 --iters 30 --warmup 5 > $GRAFT_REPO_ROOT/gpurun_out/pf_550.log 2>&1
python3 - <<PY
import csv, glob
f = glob.glob("$GRAFT_REPO_ROOT/gpurun_out/pf_550/*/*kernel_stats.csv")[0]
for r in list(csv.DictReader(open(f)))[:12]:
    print("%-70s calls %4s avg %9.1f us" % (r["Name"][:70], r["Calls"], float(r["AverageNs"]) / 1e3))
PY
