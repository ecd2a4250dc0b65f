#!/bin/bash
# sweep of the staged 3x3x3 pool kernels' tile knobs (development aid): tools/pool_sweep.sh
for dt in fp32 bf16; do for cv in 2 4 8; do for tw in 0 7; do
  echo "== $dt CV=$cv TW=$tw"
  DUALVAR_POOL_CV=$cv DUALVAR_POOL_TW=$tw python tools/pool_microbench.py --dtype $dt --only p3b,p3c,p4e,p3c_t16 2>&1 | grep -v amdgpu.ids
done; done; done
