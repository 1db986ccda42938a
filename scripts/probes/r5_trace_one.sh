# slot lengths of the key / value runs in the parity and the one-product arithmetic (library built with -DPAPR_C4_TRACE)
B=$PWD/scripts/probes/bin
mkdir -p gpurun_out/r5b
{
for one in "" 1; do
for cfg in "D_IN=117 LAYERS=5 NORM=1 DOTS=1" "D_IN=141 D_OUT=32 LAYERS=8"; do
 for mode in keep bwd; do
  echo "=== ONE=$one $cfg mode=$mode"
  env $cfg ONE=$one PAPR_HIP_LIB=$B/libpapr_trace.so SUMMARY=1 S0=0 S1=36 python3 scripts/probes/chain4_trace.py $mode 2>&1 | grep -v amdgpu.ids
 done
done
done
} > gpurun_out/r5b/trace_one.txt 2>&1
tail -5 gpurun_out/r5b/trace_one.txt
