#!/bin/bash
# run-to-run determinism of one model step (tests/h1_worker.py: forward + backward of a golden case, per-tensor gradient errors) in the default mode
for tag in lego1k chair1k; do
  for r in 1 2 3; do PAPR_WORKER_ANY_MODE=1 python3 tests/h1_worker.py $tag /tmp/det_${tag}_$r.json > /dev/null 2>&1; done
  python3 - <<PY
import json
a=[json.load(open('/tmp/det_${tag}_%d.json' % r)) for r in (1,2,3)]
for r in (1,2):
    bad=[k for k in a[0]['grads'] if a[0]['grads'][k] != a[r]['grads'][k]]
    print('$tag', 'run', r+1, 'vs 1:', 'identical' if not bad and a[0]['digest']==a[r]['digest'] and a[0]['rgb']==a[r]['rgb'] else ('rgb %r %r; differing gradient tensors: %s' % (a[0]['rgb'], a[r]['rgb'], bad[:12])))
PY
done
