#!/bin/bash
# Round 6: do the GEMM plan knobs tuned on the one-GPU step suit a SHARDED rank (7 views, one CFG half)?  tools/sim_rank.py --world 8 --ranks 1 under each setting, baseline first and last.
export TMPDIR=/tmp
run() { printf "%-44s " "$1"; env $1 python tools/sim_rank.py --world 8 --ranks 1 ${SIM_ARGS} 2>&1 | grep "^world" | sed 's/.*mixed *//'; }
run PF_NOP=0
for kv in PF_GEMM8_FILL=60 PF_GEMM8_FILL=15 PF_GEMM8_MIN_TILES=64 PF_GEMM8_MIN_TILES=256 PF_GEMM_BM128=1 PF_GEMM_BM128=0 PF_GEMM_DEEP_RING_MAX_BLOCKS=512 PF_GEMM_DEEP_RING=0 PF_GEMM_SPLIT_MINKB=6 PF_GEMM_SPLIT_MINKB=24 PF_GEMM_TAIL_SPLIT=0 PF_GEMM8_PERSIST=0 PF_GN_EPILOGUE_RES=1 "PF_ATTENTION_OCC=2" "PF_LINEAR_WS_MIN_ROWS=4096"; do run "$kv"; done
run PF_NOP=1
