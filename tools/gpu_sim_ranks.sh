#!/bin/bash
# per-rank compute time of every sharding layout on one GPU (tools/sim_rank.py); extra args: VAR=val settings to A/B
mkdir -p gpurun_out
export TMPDIR=/tmp
run() { env "$@" python tools/sim_rank.py --world 2 --ranks 0 2>&1 | grep "^world"; env "$@" python tools/sim_rank.py --world 4 --ranks 0,1 2>&1 | grep "^world"; env "$@" python tools/sim_rank.py --world 8 --ranks 0,1 2>&1 | grep "^world"; }
{ echo "== single process, same box"; python bench.py --steps 8 --warmup 2 --no-cpu-baseline 2>&1 | tail -n 1 | python -c "import sys, json; r = json.loads(sys.stdin.read()); print('world 1  %.2f ms/step' % r['ms_per_step'])"
  echo "== default"; run PF_NONE=1; for kv in "$@"; do echo "== $kv"; run $kv; done; } 2>&1 | tee gpurun_out/sim_ranks.txt
