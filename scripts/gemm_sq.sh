# SQ counters of the NT GEMM shapes (proj, qkv, down) at B=1024: where do the wave-cycles go?
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/sq
rm -rf $OUT; mkdir -p $OUT
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES -d $OUT -o sq --output-format csv -- python3 $GRAFT_REPO_ROOT/scripts/gemm_pmc.py > $OUT/log.txt 2>&1
ls -R $OUT | head -20
python3 - <<PY
import csv, glob, collections
f = glob.glob("$OUT/**/*counter_collection.csv", recursive=True)
print(f)
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for row in csv.DictReader(open(f[0])):
    k = (row["Kernel_Name"][:60], row["Grid_Size"])
    agg[k][row["Counter_Name"]] += float(row["Counter_Value"])
    cnt[(k, row["Counter_Name"])] += 1
for k, v in agg.items():
    if "gemm" not in k[0]: continue
    print(k)
    for c, x in sorted(v.items()):
        print(f"   {c:28s} {x / cnt[(k, c)]:16.0f}")
PY
