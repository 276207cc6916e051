for i in 1 2 3 4 5 6; do python -m pytest tests/test_parity_gpu.py -q -k "overflow_regrow" 2>&1 | tail -2; done
