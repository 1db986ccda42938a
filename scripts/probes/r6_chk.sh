python3 -m pytest tests/test_hip_chain_variants.py -q -m gpu -x -s -k "outgrow" 2>&1 | grep "rms error\|passed\|failed\|Error" | head -40
echo "--- with a library from before the fix (expect a failure):"
PAPR_HIP_LIB=$PWD/scripts/probes/bin/libpapr_e3.so python3 -m pytest tests/test_hip_chain_variants.py -q -m gpu -x -k "outgrow" 2>&1 | grep "AssertionError\|passed\|failed" | head -3
