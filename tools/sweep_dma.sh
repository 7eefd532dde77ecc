#!/bin/bash
# Dev tool (GPU box): forward / dgrad timings of the step's 3x3 layers under every combination of the LDS-DMA knobs.
#   bash tools/sweep_dma.sh > gpurun_out/sweep_dma.txt
for ck in 4 8; do for db in 1 0; do for tile in 256 512; do
  echo "== CK=$ck DB=$db TILE=$tile"
  TG_DMA_CK=$ck TG_DMA_DB=$db TG_DMA_TILE=$tile python tools/dma_check.py save 2>&1 | grep -v amdgpu | cut -c1-72
done; done; done
rm -f gpurun_out/dma_ref.pt
