PAPR_CHAIN=1 python scripts/probes/chain_ab.py run /tmp/a.pt
for v in slp noslp; do
  if [ $v = noslp ]; then X=-fno-slp-vectorize; else X=; fi
  bash scripts/probes/build_variant.sh dbg$v $X > /dev/null 2>&1
  for rep in 1 2 3 4; do
    PAPR_HIP_LIB=scripts/probes/bin/libpapr_dbg$v.so PAPR_C2_GENERIC=3 python scripts/probes/chain_ab.py run /tmp/c.pt 2>/dev/null | grep debug;  python scripts/probes/chain_ab.py cmp /tmp/a.pt /tmp/c.pt | grep "^d_x " | sed "s/^/DBG=$v  /"
  done
done
