# SQ / LDS counters of the row-owner kernels at B=1024 (separate passes: 8 SQ slots each)
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/rcsq
rm -rf $OUT; mkdir -p $OUT
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES -d $OUT -o a --output-format csv -- python3 $GRAFT_REPO_ROOT/scripts/rowchain_pmc.py > $OUT/log_a.txt 2>&1 || exit 1
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM_WR -d $OUT -o b --output-format csv -- python3 $GRAFT_REPO_ROOT/scripts/rowchain_pmc.py > $OUT/log_b.txt 2>&1 || exit 1
python3 - <<PY
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for f in glob.glob("$OUT/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"].split("(")[0][-46:]
        agg[k][row["Counter_Name"]] += float(row["Counter_Value"]); cnt[(k, row["Counter_Name"])] += 1
for k, v in agg.items():
    if not any(s in k for s in ("rowgemm", "mlp_")): continue
    print(k)
    for c, x in sorted(v.items()): print(f"   {c:28s} {x / cnt[(k, c)]:16.0f}")
    w = v.get("SQ_WAVE_CYCLES", 0) / max(cnt[(k, "SQ_WAVE_CYCLES")], 1)
    if w:
        f = lambda c: v.get(c, 0) / max(cnt[(k, c)], 1) / w
        print(f"   -> of a wave's life: waiting (s_waitcnt / barrier) {f('SQ_WAIT_ANY'):.2f}, waiting for issue {f('SQ_WAIT_INST_ANY'):.2f}, issuing {f('SQ_ACTIVE_INST_ANY'):.2f}; "
              f"MFMA pipe busy {v.get('SQ_VALU_MFMA_BUSY_CYCLES', 0) / max(cnt[(k, 'SQ_VALU_MFMA_BUSY_CYCLES')], 1) / max(v.get('SQ_BUSY_CYCLES', 1) / max(cnt[(k, 'SQ_BUSY_CYCLES')], 1), 1) / 4:.2f} of SIMD time; "
              f"LDS bank conflicts {v.get('SQ_LDS_BANK_CONFLICT', 0) / max(v.get('SQ_LDS_IDX_ACTIVE', 1), 1):.2f} of LDS active")
PY
