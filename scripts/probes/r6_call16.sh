#!/bin/bash
# round 6: (1) what staging / row stores cost the one-product runs (timing builds, results wrong), (2) the round's profile set on the default tree
O=gpurun_out/r6p; mkdir -p $O
run() { PAPR_BENCH_LAUNCHES=1 python3 bench.py --steps 20 --warmup 5 --no-amp-line --no-shipped-line --psnr-steps 0 --no-cpu-baseline "$@" 2> $O/l.txt | python3 -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('ms_per_step', round(j['ms_per_step'],3))"; grep "^kernel  *\(9\|10\|8\) " $O/l.txt | head -6; }
{ echo "=== amp"; run --amp
  for v in nostage nostore; do echo "=== amp $v"; PAPR_HIP_LIB=$PWD/scripts/probes/bin/libpapr_$v.so run --amp; done
  echo "=== default"; run
  for v in nostage nostore; do echo "=== default $v"; PAPR_HIP_LIB=$PWD/scripts/probes/bin/libpapr_$v.so run; done
} > $O/ablate.txt 2>&1
bash scripts/profile_round.sh r06 > $O/profile.log 2>&1
bash scripts/profile_round.sh r06amp --amp > $O/profile_amp.log 2>&1
cat $O/ablate.txt | grep "===\|ms_per"
