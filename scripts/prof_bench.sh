# rocprofv3 kernel trace + stats of the default bench (bf16, B=1024); summary CSV copied to gpurun_out/prof/
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof
rm -rf $OUT; mkdir -p $OUT
rocprofv3 --kernel-trace --stats -d $OUT -o r04 --output-format csv -- python3 $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-kernel-profile --no-extra-legs > $OUT/bench.log 2> $OUT/bench.err
ls $OUT
tail -c 600 $OUT/bench.log
