#!/bin/bash
# slot summaries (SUMMARY=1) and hot-slot stamps of the trace builds named on the command line: <tag> [mode...]
B=$PWD/scripts/probes/bin
for mode in inf keep bwd; do
  echo "=== trace  mode=$mode  (slots 0-19: length seen by waves 0 / 4, wave 0's pieces)"
  PAPR_HIP_LIB=$B/libpapr_trace.so SUMMARY=1 S0=0 S1=20 python3 scripts/probes/chain4_trace.py $mode 2>&1 | grep -v amdgpu.ids
  echo "=== trace_Knoread  mode=$mode"
  PAPR_HIP_LIB=$B/libpapr_trace_Knoread.so SUMMARY=1 S0=0 S1=20 python3 scripts/probes/chain4_trace.py $mode 2>&1 | grep -v amdgpu.ids
done
for mode in inf keep; do
  echo "=== trace_hot  mode=$mode"
  PAPR_HIP_LIB=$B/libpapr_trace_hot.so HOT=1 S0=9 S1=13 python3 scripts/probes/chain4_trace.py $mode 2>&1 | grep -v amdgpu.ids
  echo "=== trace_hot_Knoread  mode=$mode"
  PAPR_HIP_LIB=$B/libpapr_trace_hot_Knoread.so HOT=1 S0=9 S1=13 python3 scripts/probes/chain4_trace.py $mode 2>&1 | grep -v amdgpu.ids
done
