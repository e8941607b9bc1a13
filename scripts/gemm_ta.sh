cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/ta
rm -rf $OUT; mkdir -p $OUT
rocprofv3 -L 2>/dev/null | grep -E "Counter_Name" | grep -E "TA_|TCP_|TD_" | awk '{print $3}' | tr '\n' ' ' > $OUT/names.txt
cat $OUT/names.txt | head -c 3000; echo
rocprofv3 --kernel-trace --pmc TA_TA_BUSY_sum TA_BUSY_avr TCP_PENDING_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum TCP_GATE_EN1_sum TCP_GATE_EN2_sum -d $OUT -o ta --output-format csv -- python3 $GRAFT_REPO_ROOT/scripts/gemm_pmc.py > $OUT/log.txt 2>&1
python3 - <<PY
import csv, glob, collections
f = glob.glob("$OUT/**/*counter_collection.csv", recursive=True)
if not f: print(open("$OUT/log.txt").read()[-1500:]); raise SystemExit
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for row in csv.DictReader(open(f[0])):
    k = (row["Kernel_Name"][:60], row["Grid_Size"])
    agg[k][row["Counter_Name"]] += float(row["Counter_Value"]); cnt[(k, row["Counter_Name"])] += 1
for k, v in agg.items():
    if "gemm" not in k[0]: continue
    print(k)
    for c, x in sorted(v.items()): print(f"   {c:34s} {x / cnt[(k, c)]:16.0f}")
PY
