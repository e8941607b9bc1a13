cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/lds
rm -rf $OUT; mkdir -p $OUT
rocprofv3 -L 2>/dev/null | grep -i -E "LDS" | head -40 > $OUT/counters.txt
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SMEM SQ_ACTIVE_INST_VALU SQ_WAVES -d $OUT -o lds --output-format csv -- python3 $GRAFT_REPO_ROOT/scripts/gemm_pmc.py > $OUT/log.txt 2>&1
python3 - <<PY
import csv, glob, collections
f = glob.glob("$OUT/**/*counter_collection.csv", recursive=True)
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for row in csv.DictReader(open(f[0])):
    k = (row["Kernel_Name"][:60], row["Grid_Size"])
    agg[k][row["Counter_Name"]] += float(row["Counter_Value"]); cnt[(k, row["Counter_Name"])] += 1
for k, v in agg.items():
    if "gemm" not in k[0]: continue
    print(k)
    for c, x in sorted(v.items()): print(f"   {c:28s} {x / cnt[(k, c)]:16.0f}")
PY
head -30 $OUT/counters.txt; tail -3 $OUT/log.txt
