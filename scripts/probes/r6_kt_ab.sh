#!/bin/bash
# kernel traces of the default step with / without the tail kernel's f16 gradient rows:   bash scripts/probes/r6_kt_ab.sh [bench args]
O=gpurun_out/r6kt; mkdir -p $O; export TMPDIR=/tmp
for v in 1 0; do
  export PAPR_TAIL_F16=$v
  rocprofv3 --kernel-trace -d $O/kt$v -o kt -- python3 bench.py --steps 10 --warmup 3 --profile-steps 0 --no-cpu-baseline --no-amp-line --no-shipped-line --psnr-steps 0 "$@" > $O/bench$v.json 2> $O/kt$v.err
  python3 scripts/rocpd_summary.py $(find $O/kt$v -name "*.db" | head -1) last:10 > $O/kernel_trace_$v.txt
done
find $O -name "*.db" -delete
for v in 1 0; do echo "== PAPR_TAIL_F16=$v"; head -14 $O/kernel_trace_$v.txt | cut -c1-160; done
