#!/bin/bash
# two processes, the same steps: where do their gradients first differ?   bash scripts/probes/r6_det_train.sh nerfsyn/lego.yml 12
scene=${1:-nerfsyn/lego.yml}; steps=${2:-12}
python3 scripts/probes/r6_det_train.py $scene $steps /tmp/dt1.pt > /tmp/dt1.log 2>&1 || tail -5 /tmp/dt1.log
python3 scripts/probes/r6_det_train.py $scene $steps /tmp/dt2.pt > /tmp/dt2.log 2>&1 || tail -5 /tmp/dt2.log
python3 - <<'PY'
import torch
a, b = torch.load('/tmp/dt1.pt'), torch.load('/tmp/dt2.pt')
for s, (x, y) in enumerate(zip(a, b)):
    bad = [k for k in x if x[k] != y[k]]
    print('step', s, 'identical' if not bad else 'DIFFER: %d of %d entries: %s' % (len(bad), len(x), bad[:8]))
    if bad:
        for k in bad[:8]: print('   ', k, x[k], y[k])
        break
PY
