#!/bin/bash
# runs the panel-GEMM probe (tools/probes/panel_gemm.hip, built in-tree beforehand) on the decoder layer's shapes
cd "$(dirname "$0")/probes" || exit 1
P=${1:-./panel_gemm}
SHAPES=${SHAPES:-"5120,512,512,20,100 5120,512,512,20,50 2560,512,512,20,100 5120,1536,512,20,50 2560,1536,512,20,100 5120,2048,512,20,50 5120,512,2048,20,50 5120,512,1536,20,50 14592,512,512,20,60 14592,2048,512,20,26 14592,512,2048,20,26 1280,512,512,20,58 7680,1024,512,20,100"}
for shp in $SHAPES; do
  timeout 120 $P ${shp//,/ } || echo "FAILED: $shp"
done
