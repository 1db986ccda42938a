# kernel trace of the bench step only (the full set: scripts/profile_round.sh)
OUT=gpurun_out/r5b/kt_$1; shift
mkdir -p $OUT
export TMPDIR=/tmp
rocprofv3 --kernel-trace -d $OUT/kt -o kt -- python3 bench.py --steps 10 --warmup 3 --profile-steps 0 --no-cpu-baseline --no-amp-line --no-shipped-line --psnr-steps 0 "$@" > $OUT/bench_under_trace.json 2> $OUT/kt.err
python3 scripts/rocpd_summary.py $(find $OUT/kt -name "*.db" | head -1) last:10 > $OUT/kernel_trace.txt
find $OUT -name "*.db" -delete
cat $OUT/kernel_trace.txt | cut -c1-160
