cd /root/repo
python -m pytest tests/test_gemm_modes_gpu.py -m gpu -q -x 2>&1 | tail -5
for m in bf16x3 bf16; do
NACF_GEMM_MODE=$m python tools/bf16_trace.py 15360:1024:512 128 --images | grep -v "outside"
done
python tools/gemm_bench.py --modes bf16x3,bf16 --images
