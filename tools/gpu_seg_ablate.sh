#!/bin/bash
# Round 6: what the per-tap source-offset recompute (set_segment) of the conv kernels costs -- sustained loops (3000 launches per shape), same box:
#   general form (PF_CONV_FASTSEG=0) vs the packed form (default) [vs a timing-only build that skips it: PF_HIP_LIB=panfusion_amd/abl/lib_abl_NOSEG.so, if present]
export TMPDIR=/tmp
S=${SHAPES:-conv64,conv64cat,conv32,conv16,conv8,conv64res,pano_conv64,pano_conv32}
for rep in 1 2; do
  echo "## general form (PF_CONV_FASTSEG=0)"; PF_CONV_FASTSEG=0 python tools/gemm_bench.py --reps 3000 --shapes $S 2>&1 | grep TF/s
  echo "## packed form (default)"; python tools/gemm_bench.py --reps 3000 --shapes $S 2>&1 | grep TF/s
  if [ -f panfusion_amd/abl/lib_abl_NOSEG.so ]; then echo "## timing-only: offsets of tap 0 for every tap"; PF_HIP_LIB=panfusion_amd/abl/lib_abl_NOSEG.so python tools/gemm_bench.py --reps 3000 --shapes $S 2>&1 | grep TF/s; fi
done
