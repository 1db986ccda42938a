# A/B of chain2 build variants on the GPU box: VARIANTS="tag[:flag[:flag]] ..." (built here beforehand with build_variant.sh)
for v in ${VARIANTS:-base}; do
  tag=${v%%:*}
  echo "== $v"; PAPR_HIP_LIB=scripts/probes/bin/libpapr_$tag.so PAPR_CHAIN=2 python scripts/probes/chain_bench.py 2>/dev/null
  if [ "$CMP" = "1" ]; then
    PAPR_CHAIN=1 python scripts/probes/chain_ab.py run /tmp/a.pt > /dev/null 2>&1; PAPR_HIP_LIB=scripts/probes/bin/libpapr_$tag.so python scripts/probes/chain_ab.py run /tmp/c.pt >/dev/null 2>&1 && python scripts/probes/chain_ab.py cmp /tmp/a.pt /tmp/c.pt | grep -v " mismatched 0 " | head -3
  fi
done
