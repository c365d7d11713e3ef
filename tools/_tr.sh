cd /root/repo
python -m pytest tests/test_gemm_modes_gpu.py -m gpu -q -x 2>&1 | tail -3
NACF_GEMM_MODE=bf16x3 python tools/bf16_trace.py 15360:1024:512 128 --images | grep -v "outside"
python tools/gemm_bench.py --modes bf16x3 --tiles 128 --images
