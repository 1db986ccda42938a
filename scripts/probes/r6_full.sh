#!/bin/bash
# the whole GPU suite, smoke() and the default bench line on one box
O=gpurun_out/r6full; mkdir -p $O
timeout 3000 python3 -m pytest tests -q -m gpu -x 2>&1 | tail -8 > $O/tests.txt
timeout 300 python3 -c "import __graft_entry__ as g; g.smoke(); print('smoke OK')" 2>&1 | tail -2 > $O/smoke.txt
timeout 900 python3 bench.py > $O/bench.json 2> $O/bench.err
cat $O/tests.txt $O/smoke.txt; python3 -c "
import json; d=json.loads(open('$O/bench.json').read().strip().splitlines()[-1]); print(d['ms_per_step'], d['value'], [(k, v.get('ms_per_step')) for k,v in d.items() if isinstance(v, dict) and 'ms_per_step' in v])"
