#!/bin/bash
# bench.py in a second checkout of the repository (./_old: `git worktree add _old <commit>`, built there) against this one, alternating, in one box
run() { (cd $1 && python3 bench.py --steps 20 --warmup 5 --no-amp-line --no-shipped-line --psnr-steps 0 --no-cpu-baseline --profile-steps 0 2>/dev/null | python3 -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$2', round(j['ms_per_step'],3))"); }
for i in 1 2 3; do run _old old; run . new; done
