cd /root/repo
timeout 600 python -m pytest tests/test_gemm_modes_gpu.py -x -q 2>&1 | tail -3
timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -5
