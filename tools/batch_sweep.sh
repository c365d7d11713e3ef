for b in 64 128 256 512 1024; do
  python bench.py --no-cpu-baseline --no-decode --no-compare --no-loader --batch $b --steps 20 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['config']['global_batch'], d['value'], d['ms_per_step'], d['roofline']['all_gemm_tflops'], d['roofline']['frac'])"
done
