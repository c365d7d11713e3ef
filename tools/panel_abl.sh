#!/bin/bash
cd "$(dirname "$0")/probes" || exit 1
for abl in 0 1 2 3 4 7 15; do
  for shp in "5120 1536 512 10 50" "5120 512 2048 10 50"; do
    timeout 120 ./panel_gemm_abl$abl $shp | grep -E "panel 64x128|per tile|by position"
  done
done
