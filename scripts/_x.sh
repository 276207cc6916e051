timeout 1500 python -m pytest tests -m gpu -q -x 2>&1 | tail -2
python scripts/fuzz_parity.py 4000 343401 2>&1 | tail -1
FUZZ_BIG=1 python scripts/fuzz_parity.py 150 343402 2>&1 | tail -1
python scripts/fuzz_features.py 1500 343403 2>&1 | tail -1
