# kernel trace of the 800 x 800 render (scripts/bench_render.py): per-kernel summary of the whole run
OUT=gpurun_out/r5b/rt_$1; shift
mkdir -p $OUT
export TMPDIR=/tmp
rocprofv3 --kernel-trace -d $OUT/kt -o kt -- python3 scripts/bench_render.py 30000 400 > $OUT/render.txt 2> $OUT/kt.err
python3 scripts/rocpd_summary.py $(find $OUT/kt -name "*.db" | head -1) 0.5 > $OUT/kernel_trace.txt
find $OUT -name "*.db" -delete
cat $OUT/render.txt | grep render; head -30 $OUT/kernel_trace.txt | cut -c1-150
