# HBM traffic per kernel family of the default bench step (bf16, B=1024): rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in
# SEPARATE passes (TCC slots), kernel-trace only.  gfx950 corrections per MI355X_MICROARCH.md (HBM section): FETCH_SIZE x2 for
# wide coalesced reads; both counters are in KiB.  Writes gpurun_out/pmc/pmc_traffic{,_detail}.json (copy into profiles/).
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc
rm -rf $OUT; mkdir -p $OUT
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $OUT -o fetch --output-format csv -- python3 $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 2 --no-cpu-baseline --no-kernel-profile --no-extra-legs > $OUT/fetch.log 2>&1 || exit 1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $OUT -o write --output-format csv -- python3 $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 2 --no-cpu-baseline --no-kernel-profile --no-extra-legs > $OUT/write.log 2>&1 || exit 1
python3 - <<PY
import csv, glob, json, collections
FAM = [("rowgemm", "mmfm_rowgemm"), ("mlp_fwd", "mmfm_mlp_fwd"), ("mlp_bwd", "mmfm_mlp_bwd"), ("ln_linear_grad", "mmfm_ln_linear_grad"),
       ("prep_weights", "mmfm_prep_weights"), ("gemm_bf16_kernel", "mmfm_gemm"), ("gemm_dw_kernel", "mmfm_gemm"), ("gemm_big_kernel", "mmfm_gemm"), ("gemm_f32_kernel", "mmfm_gemm"), ("attn_bwd", "mmfm_attn_bwd"), ("attn_fwd", "mmfm_attn_fwd"), ("attn_keepbits", "mmfm_attn_keepbits"),
       ("ln_bwd_kernel", "mmfm_layernorm_bwd"), ("ln_fwd_kernel", "mmfm_layernorm_fwd"), ("reduce_slabs", "mmfm_reduce_slabs"),
       ("adamw_kernel", "mmfm_adamw_step"), ("loss_", "mmfm_masked_loss"), ("stitch_", "mmfm_stitch"), ("onehot_kernel", "mmfm_stitch"),
       ("dropout_apply", "mmfm_dropout_apply")]
def fam(name):
    for k, f in FAM:
        if k in name: return f
    return None
def collect(prefix, counter):
    tot, cnt = collections.Counter(), collections.Counter()
    f = glob.glob("$OUT/**/%s_counter_collection.csv" % prefix, recursive=True)[0]
    for row in csv.DictReader(open(f)):
        if row["Counter_Name"] != counter: continue
        k = fam(row["Kernel_Name"])
        if k: tot[k] += float(row["Counter_Value"]); cnt[k] += 1
    return tot, cnt
ft, fc = collect("fetch", "FETCH_SIZE")
wt, wc = collect("write", "WRITE_SIZE")
detail, flat = {}, {}
for k in sorted(set(ft) | set(wt)):
    fb = 2.0 * 1024.0 * ft[k] / max(fc[k], 1)
    wb = 1024.0 * wt[k] / max(wc[k], 1)
    detail[k] = dict(hbm_bytes_per_launch=int(fb + wb), fetch_bytes_per_launch=int(fb), write_bytes_per_launch=int(wb), launches_sampled=int(fc[k]),
                     note="rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes, B=1024 bf16 bench step; FETCH_SIZE x2 (gfx950 wide-read correction), KiB->B")
    flat[k] = int(fb + wb)
import importlib.util, os
spec = importlib.util.spec_from_file_location("bench", os.path.join(os.environ["GRAFT_REPO_ROOT"], "bench.py")); bench = importlib.util.module_from_spec(spec); spec.loader.exec_module(bench)
flat["src_sha"] = bench.src_sha()          # bench.py attaches these figures to a run only if the kernel sources are the same
json.dump(flat, open("$OUT/pmc_traffic.json", "w"), indent=1)
json.dump(detail, open("$OUT/pmc_traffic_detail.json", "w"), indent=1)
print(json.dumps(flat, indent=1))
PY
