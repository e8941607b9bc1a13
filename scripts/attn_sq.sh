# SQ counters of the bf16 attention kernels at B=1024: issue-bound or stall-bound?   usage: attn_sq.sh [tag] (env MMFM_ATTN_FAST / MMFM_ATTN_MASK select the kernels)
cd /tmp && export TMPDIR=/tmp
TAG=${1:-sqa}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
rm -rf $OUT; mkdir -p $OUT
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES -d $OUT -o sq --output-format csv -- python3 $GRAFT_REPO_ROOT/scripts/attn_bench.py 1024 0.4 3 > $OUT/log.txt 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_ACTIVE_INST_VALU SQ_WAVES -d $OUT -o sq2 --output-format csv -- python3 $GRAFT_REPO_ROOT/scripts/attn_bench.py 1024 0.4 3 > $OUT/log2.txt 2>&1
rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_SALU SQ_WAIT_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE -d $OUT -o sq3 --output-format csv -- python3 $GRAFT_REPO_ROOT/scripts/attn_bench.py 1024 0.4 3 > $OUT/log3.txt 2>&1
python3 - <<PY > $OUT/summary.txt
import csv, glob, collections
for f in sorted(glob.glob("$OUT/**/*counter_collection.csv", recursive=True)):
    agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
    for row in csv.DictReader(open(f)):
        k = (row["Kernel_Name"][:70], row["Grid_Size"], row["LDS_Block_Size"], row["VGPR_Count"], row.get("Accum_VGPR_Count"), row.get("SGPR_Count"))
        agg[k][row["Counter_Name"]] += float(row["Counter_Value"])
        cnt[(k, row["Counter_Name"])] += 1
    for k, v in agg.items():
        if "attn" not in k[0]: continue
        print(k)
        for c, x in sorted(v.items()):
            print(f"   {c:28s} {x / cnt[(k, c)]:16.0f}")
for f in sorted(glob.glob("$OUT/**/sq_kernel_trace.csv", recursive=True)):
    d = collections.defaultdict(list)
    for row in csv.DictReader(open(f)):
        d[row["Kernel_Name"][:70]].append((int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) / 1e3)
    for k, v in d.items():
        if "attn" in k: print(k, "n", len(v), "avg us %.1f" % (sum(v) / len(v)))
PY
cat $OUT/summary.txt
tail -2 $OUT/log.txt
