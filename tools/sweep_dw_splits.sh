# dW (kind 2) / dX (kind 1) tile x split sweep on the shapes of one NACF train step (tuning aid)
for shp in 2:2980:512:2048 2:2980:2048:512 2:2980:1536:512 2:2980:512:512 2:7680:1024:512 2:7680:512:2048 2:15360:1024:512 2:1490:10547:512; do
  for s in ${SPLITS:-1 2 4 8 16}; do
    echo -n "$shp splits=$s  "
    NACF_GEMM_SPLITS=$s timeout 100 python tools/gemm_bench.py --iters 20 --shapes $shp 2>&1 | grep custom | awk '{print $4, $5, $6, $7, $8}'
  done
done
