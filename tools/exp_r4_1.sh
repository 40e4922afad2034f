cd /root/repo
python tools/bench_resjac.py 4096 2>&1 | tail -3
python tools/bench_b64.py 64 128 2>&1 | tail -12
python -m pytest tests/test_parity_gpu.py -x -q -m gpu 2>&1 | tail -3
