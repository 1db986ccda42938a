#!/bin/bash
# kernel trace of the default step (and with --amp):   bash scripts/probes/r6_kt.sh
O=gpurun_out/r6kt; mkdir -p $O; export TMPDIR=/tmp
for v in default amp; do
  a=""; [ $v = amp ] && a="--amp"
  rocprofv3 --kernel-trace -d $O/kt_$v -o kt -- python3 bench.py --steps 10 --warmup 3 --profile-steps 0 --no-cpu-baseline --no-amp-line --no-shipped-line --psnr-steps 0 $a > $O/bench_$v.json 2> $O/kt_$v.err
  python3 scripts/rocpd_summary.py $(find $O/kt_$v -name "*.db" | head -1) last:10 > $O/kernel_trace_$v.txt
done
find $O -name "*.db" -delete
for v in default amp; do echo "== $v"; head -16 $O/kernel_trace_$v.txt | cut -c1-160; done
