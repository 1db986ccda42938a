for S in ${SETS}; do
  echo "=== $S"
  if [ "$S" = none ]; then E=""; else E="${S//,/ }"; fi
  env $E python3 scripts/bench_render.py 30000 400 2>&1 | grep render
done
