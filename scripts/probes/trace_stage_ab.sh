#!/bin/bash
# fine stamps of the two staging slots of a pair of tiles (slots 7, 8) for variant builds of chain4.hip:  bash scripts/probes/trace_stage_ab.sh <tag>=<flags> ...
for a in "$@"; do
  tag=${a%%=*}; flags=${a#*=}
  bash scripts/probes/c4_variant.sh $tag -DPAPR_C4_TRACE -DPAPR_C4_TRACE_FINE $flags > /dev/null 2>&1
  echo "##### $tag: $flags"
  bash scripts/probes/trace_fine.sh $tag | grep -v "^  [ 0-9] \|^  wave"
done
