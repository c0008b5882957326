python tests/fuzz_parity.py 2500 909090 2>&1 | tail -1
python tests/fuzz_parity.py 3000 808080 chunked 2>&1 | tail -1
python tests/fuzz_parity.py 300 707070 training 2>&1 | tail -2
