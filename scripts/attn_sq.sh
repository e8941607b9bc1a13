# SQ counters of the bf16 attention kernels at B=1024: issue-bound or stall-bound?
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/sqa
rm -rf $OUT; mkdir -p $OUT
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES -d $OUT -o sq --output-format csv -- python3 $GRAFT_REPO_ROOT/scripts/attn_bench.py 1024 0.4 3 > $OUT/log.txt 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_ACTIVE_INST_VALU SQ_WAVES -d $OUT -o sq2 --output-format csv -- python3 $GRAFT_REPO_ROOT/scripts/attn_bench.py 1024 0.4 3 > $OUT/log2.txt 2>&1
python3 - <<PY
import csv, glob, collections
for f in sorted(glob.glob("$OUT/**/*counter_collection.csv", recursive=True)):
    agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
    for row in csv.DictReader(open(f)):
        k = (row["Kernel_Name"][:70], row["Grid_Size"], row["LDS_Block_Size"], row["VGPR_Count"], row.get("Accum_VGPR_Count"))
        agg[k][row["Counter_Name"]] += float(row["Counter_Value"])
        cnt[(k, row["Counter_Name"])] += 1
    for k, v in agg.items():
        if "attn" not in k[0]: continue
        print(k)
        for c, x in sorted(v.items()):
            print(f"   {c:28s} {x / cnt[(k, c)]:16.0f}")
PY
cat $OUT/log.txt | tail -3
