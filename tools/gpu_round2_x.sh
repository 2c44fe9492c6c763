timeout 300 python -m pytest tests/test_gpu_parity.py -m gpu -q -k "table_driven or sox_golden" 2>&1 | grep -E "^E  *(Assert|assert)|passed|failed" | head
