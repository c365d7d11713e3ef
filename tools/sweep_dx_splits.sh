for shp in 1:2980:2048:512 1:2980:1536:512 1:2980:512:512 1:2980:512:2048 1:7680:1024:512 1:15360:1024:512 1:1490:10547:512 1:821:10547:512; do
  for s in 1 2 3 4 8; do
    echo -n "$shp splits=$s  "
    NACF_GEMM_SPLITS=$s timeout 100 python tools/gemm_bench.py --iters 20 --shapes $shp 2>&1 | grep custom | awk '{print $4, $5, $6, $7, $8}'
  done
done
