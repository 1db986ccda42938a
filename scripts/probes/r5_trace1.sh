B=$PWD/scripts/probes/bin
mkdir -p gpurun_out/r5b
{
for cfg in "D_IN=117 LAYERS=5 NORM=1 DOTS=1" "D_IN=141 D_OUT=32 LAYERS=8"; do
 for mode in keep bwd inf; do
  echo "=== $cfg mode=$mode"
  env $cfg PAPR_HIP_LIB=$B/libpapr_trace.so SUMMARY=1 S0=0 S1=36 python3 scripts/probes/chain4_trace.py $mode 2>&1 | grep -v amdgpu.ids
 done
done
} > gpurun_out/r5b/trace_summary.txt 2>&1
{
for cfg in "D_IN=117 LAYERS=5 NORM=1 DOTS=1" "D_IN=141 D_OUT=32 LAYERS=8"; do
 for mode in keep bwd; do
  echo "=== FINE $cfg mode=$mode"
  env $cfg PAPR_HIP_LIB=$B/libpapr_trace_fine.so STAMPS=7 S0=8 S1=24 python3 scripts/probes/chain4_trace.py $mode 2>&1 | grep -v amdgpu.ids
 done
done
} > gpurun_out/r5b/trace_fine.txt 2>&1
tail -3 gpurun_out/r5b/trace_fine.txt
