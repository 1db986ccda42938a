B=$PWD/scripts/probes/bin
for raw in "" 1; do
  echo "=== FINE key keep RAW=$raw"
  RAW=$raw D_IN=117 LAYERS=5 NORM=1 DOTS=1 PAPR_HIP_LIB=$B/libpapr_trace_fine.so STAMPS=7 S0=18 S1=22 python3 scripts/probes/chain4_trace.py keep 2>&1 | grep -v amdgpu.ids
done
