V=$PWD/gpuspectral_amd/lib/variants
GSP_LIB_PATH=$V/w8_6.so timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "deep_tree" 2>&1 | grep -E "assert|Error|passed|failed" | head
