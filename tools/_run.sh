cd /root/repo
python -m pytest tests/test_gemm_modes_gpu.py -m gpu -q -x 2>&1 | tail -3
for m in bf16x3 bf16; do echo "== $m"; python tools/step_gemm_breakdown.py $m 2>&1 | grep -v amdgpu.ids; done
for m in bf16x3 bf16; do
python bench.py --gemm-mode $m --no-cpu-baseline --no-compare --no-loader --steps 50 2>&1 | tail -1 > gpurun_out/r2e/bench_$m.json
python - <<PY
import json
d=json.load(open("gpurun_out/r2e/bench_$m.json"))
print("$m", d["value"], d["ms_per_step"], d["roofline"]["all_gemm_ms_per_step"], d["roofline"]["all_gemm_tflops"], d["decode"]["captions_per_s"], d["final_loss"])
PY
done
