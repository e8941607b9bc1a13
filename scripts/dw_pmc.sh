# HBM fetch bytes and L2 hit rate of the dW kernels (scripts/dw_bench.py), separate --pmc passes.  Prints per-kernel averages.
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/dwpmc
rm -rf $OUT; mkdir -p $OUT
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $OUT -o fetch --output-format csv -- python3 $GRAFT_REPO_ROOT/scripts/dw_bench.py > $OUT/fetch.log 2>&1 || exit 1
rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum -d $OUT -o hit --output-format csv -- python3 $GRAFT_REPO_ROOT/scripts/dw_bench.py > $OUT/hit.log 2>&1 || exit 1
python3 - <<PY
import csv, glob, collections
for prefix in ("fetch", "hit"):
    f = glob.glob("$OUT/**/%s_counter_collection.csv" % prefix, recursive=True)[0]
    acc = collections.defaultdict(list)
    for row in csv.DictReader(open(f)):
        if "gemm" not in row["Kernel_Name"]: continue
        acc[(row["Kernel_Name"][:40], row["Grid_Size"], row["Counter_Name"])].append(float(row["Counter_Value"]))
    for k, v in acc.items():
        print(prefix, k, "n", len(v), "avg", sum(v) / len(v))
PY
