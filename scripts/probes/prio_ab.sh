for v in base "free -DC2_FREE"; do
  set -- $v; tag=$1; shift
  bash scripts/probes/build_variant.sh $tag "$@" > /dev/null 2>&1
  echo "== $tag $@"; PAPR_HIP_LIB=scripts/probes/bin/libpapr_$tag.so PAPR_CHAIN=2 python scripts/probes/chain_bench.py 2>/dev/null
  PAPR_CHAIN=1 python scripts/probes/chain_ab.py run /tmp/a.pt > /dev/null 2>&1; PAPR_HIP_LIB=scripts/probes/bin/libpapr_$tag.so python scripts/probes/chain_ab.py run /tmp/c.pt >/dev/null 2>&1 && python scripts/probes/chain_ab.py cmp /tmp/a.pt /tmp/c.pt | grep -v " mismatched 0 " | head -3
done
