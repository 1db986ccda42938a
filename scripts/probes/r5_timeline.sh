# kernel timeline of the last bench step (scripts/rocpd_timeline.py): every launch with its gap
OUT=gpurun_out/r5b/tl_$1; shift
mkdir -p $OUT
export TMPDIR=/tmp
rocprofv3 --kernel-trace -d $OUT/kt -o kt -- python3 bench.py --steps 6 --warmup 3 --profile-steps 0 --no-cpu-baseline --no-amp-line --no-shipped-line --psnr-steps 0 "$@" > $OUT/bench_under_trace.json 2> $OUT/kt.err
DB=$(find $OUT/kt -name "*.db" | head -1)
python3 scripts/rocpd_timeline.py $DB > $OUT/timeline.txt
python3 scripts/rocpd_summary.py $DB last:5 > $OUT/kernel_trace.txt
find $OUT -name "*.db" -delete
tail -3 $OUT/timeline.txt
