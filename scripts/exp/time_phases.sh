#!/bin/bash
# per-phase cost of deepfm_fwd_bwd_kernel: run the standalone timing with the kernel cut after phase N
for s in 1 2 3 4 5 6 0; do
  echo -n "stop=$s  "; REC_FUSED_STOP=$s python scripts/exp/time_fused.py 2>/dev/null | grep fwd_bwd
done
