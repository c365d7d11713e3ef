cd /root/repo
for m in bf16x3 bf16; do
echo "=== $m dW"
NACF_GEMM_MODE=$m SPLITS=0,2,4,8,16,32 python tools/dw_rows_bench.py 2:5120:2980:512:2048,2:5120:2980:2048:512,2:5120:2980:512:512,2:5120:2980:1536:512,2:5120:2311:10547:512 2>&1 | grep -v amdgpu
echo "=== $m fwd/dX"
NACF_GEMM_MODE=$m SPLITS=0 python tools/dw_rows_bench.py 0:5120:2980:1536:512,0:5120:2980:512:512,0:5120:2980:2048:512,0:5120:2980:512:2048,1:5120:2980:512:2048,1:5120:2980:2048:512,1:5120:2980:512:512,1:5120:2980:1536:512 2>&1 | grep -v amdgpu
NACF_GEMM_MODE=$m SPLITS=0,2,4,8 python tools/dw_rows_bench.py 1:5120:2311:10547:512 2>&1 | grep -v amdgpu
done
