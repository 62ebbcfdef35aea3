#!/bin/bash
# The ONE parameterised GPU-box runner (replaces the per-pass gpu_r*.sh / gpu_final*.sh / gpu_*_ab.sh scripts).
#
#   gpurun --timeout 1500 -- 'bash tools/gpu_run.sh <tag> <section> [<section> ...]'
#
# Every section writes gpurun_out/<tag>_*; copy what is to be judged into profiles/.  Sections:
#   tests[:<pytest -k expr>]   the -m gpu suite (optionally filtered)          -> <tag>_pytest_gpu.log
#   envtests:"VAR=v":<-k expr> the (filtered) suite under environment switches         -> <tag>_pytest_gpu_<env>.log
#   gemmenv:"VAR=v;VAR2=v"     GEMM microbenchmark under each setting (baseline first)  -> <tag>_gemm_env.txt
#   attnenv:"VAR=v;VAR2=v"     attention microbenchmark under each setting (baseline first) -> <tag>_attn_env.txt
#   smoke                      __graft_entry__.smoke()                         -> <tag>_smoke.log
#   bench[:<bench.py args>]    one bench line (+ per-shape table)              -> <tag>_bench[_<args>].json, <tag>_shapes.txt
#   env:"VAR=v VAR2=v;VAR3=v"  same-box A/B (settings separated by ;) of environment switches on the default bench (baseline first and last; the two
#                              baseline lines ARE the noise floor of that box)  -> <tag>_env_ab.txt
#   lib:<path.so>              same-box A/B of library builds (PF_HIP_LIB) on the default bench -> <tag>_lib_ab.txt
#   stats                      rocprofv3 --kernel-trace --stats of the default bench command -> <tag>_kernel_stats_default_cmd.csv
#   serial[:<bench args>]      one stream, no graphs: per-kernel table        -> <tag>_kernels_serial.txt
#   pmc[:<bench args>]         FETCH_SIZE / WRITE_SIZE passes (separate, --kernel-trace only) -> <tag>_pmc_*.txt, <tag>_traffic.json
#   streams                    eager two-stream critical path                  -> <tag>_streams.txt
#   mfma                       MFMA-pipe busy % of every dispatch in the step  -> <tag>_mfma_instep.txt
#   gemm[:<shapes>] attn elem vae   kernel microbenchmarks (gemm: + --phases stamps with gemmphases[:shapes])
#   train                      training-step benches (LoRA + EPA; layout-conditioned)
#   sim[:"VAR=v"]              per-rank compute time of the sharded layouts on one GPU (tools/sim_rank.py)
#   simprof[:"--world 8 --ranks 1"]  rocprofv3 per-kernel table of one simulated rank      -> <tag>_simprof_*.txt
#   dist[:"2 4 8"]             functional dry run of the sharded path over gloo on one GPU (tools/gpu_dist_dry.sh)
mkdir -p gpurun_out
export TMPDIR=/tmp
ulimit -c 0
R=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=${1:-run}; shift
NOLEGS="--no-cpu-baseline --no-training-leg"
q() { grep -v amdgpu.ids; }
line() { tail -n 1 "$1" | python -c "
import sys, json
try:
    r = json.loads(sys.stdin.read()); rf = r.get('roofline') or {}
    print('%-40s %6.2f steps/s %7.2f ms | gemm %4.0f TF/s frac %.3f' % ('$2', r['value'], r['ms_per_step'], rf.get('achieved', 0), rf.get('frac', 0)))
except Exception as e:
    print('$2: bench failed', e)
"; }
slug() { echo "$1" | tr -c 'A-Za-z0-9=\n' '_' | cut -c1-60; }

for SEC in "$@"; do
  NAME=${SEC%%:*}; ARG=""; [ "$SEC" != "$NAME" ] && ARG=${SEC#*:}
  echo "== $NAME $ARG"
  cd $R
  case $NAME in
    tests)
      timeout 1700 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider -n 4 ${ARG:+-k "$ARG"} 2>&1 | q | tail -n 40 | cut -c1-300 > gpurun_out/${TAG}_pytest_gpu.log
      tail -n 6 gpurun_out/${TAG}_pytest_gpu.log ;;
    envtests)    # envtests:"VAR=v VAR2=v":<pytest -k expr>  -- the (filtered) suite under environment switches (A/B kernels behind a switch)
      EV=${ARG%%:*}; KX=""; [ "$ARG" != "$EV" ] && KX=${ARG#*:}
      env $EV timeout 1700 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider -n 4 ${KX:+-k "$KX"} 2>&1 | q | tail -n 40 | cut -c1-300 > gpurun_out/${TAG}_pytest_gpu_$(slug "$EV").log
      tail -n 6 gpurun_out/${TAG}_pytest_gpu_$(slug "$EV").log ;;
    gemmenv)     # gemmenv:"VAR=v;VAR2=v"  -- GEMM microbenchmark (16-bit and fp32-stream epilogues) under each setting, baseline first
      { IFS=';' read -ra KVS <<< "$ARG"
        for kv in "PF_NOP=0" "${KVS[@]}"; do
          echo "## $kv"; env $kv python tools/gemm_bench.py --reps 20 $GEMM_ARGS 2>&1 | q
          echo "## $kv --stream32"; env $kv python tools/gemm_bench.py --reps 20 --stream32 --shapes ${GEMM_SHAPES32:-lin320,ff2_320,lin640,lin1280,lin_mid} 2>&1 | q
        done; } | tee gpurun_out/${TAG}_gemm_env.txt | tail -n 90 ;;
    attnenv)     # attnenv:"VAR=v;VAR2=v"  -- attention microbenchmark under each setting, baseline first
      { IFS=';' read -ra KVS <<< "$ARG"
        for kv in "PF_NOP=0" "${KVS[@]}"; do echo "## $kv"; env $kv python tools/attn_bench.py $ATTN_ARGS 2>&1 | q; done; } | tee gpurun_out/${TAG}_attn_env.txt ;;
    smoke)
      timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | q | tail -n 3 | tee gpurun_out/${TAG}_smoke.log ;;
    bench)
      S=$(slug "$ARG"); F=gpurun_out/${TAG}_bench${S:+_$S}
      timeout 900 python bench.py $ARG --trace-out gpurun_out/${TAG}_shapes${S:+_$S}.txt > $F.log 2>&1
      tail -n 1 $F.log > $F.json; line $F.json "bench $ARG" ;;
    env)
      { run() { env $1 python bench.py $NOLEGS --steps 10 --warmup 2 2>&1 | tail -n 1 > /tmp/l.json; line /tmp/l.json "$1"; }
        IFS=';' read -ra KVS <<< "$ARG"
        run "PF_NOP=0"; for kv in "${KVS[@]}"; do run "$kv"; done; run "PF_NOP=1"; } 2>&1 | tee -a gpurun_out/${TAG}_env_ab.txt ;;
    lib)
      { for lib in "" "$ARG" ""; do PF_HIP_LIB=$lib python bench.py $NOLEGS --steps 10 --warmup 2 2>&1 | tail -n 1 > /tmp/l.json; line /tmp/l.json "lib=${lib:-current}"; done; } 2>&1 | tee -a gpurun_out/${TAG}_lib_ab.txt ;;
    stats)
      cd /tmp
      timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${TAG}_g -o bench -- python $R/bench.py $NOLEGS > $R/gpurun_out/${TAG}_rocprof_graphs.log 2>&1
      cp $(find $R/gpurun_out/${TAG}_g -name '*kernel_stats.csv' | head -1) $R/gpurun_out/${TAG}_kernel_stats_default_cmd.csv 2>/dev/null
      rm -rf $R/gpurun_out/${TAG}_g
      head -n 8 $R/gpurun_out/${TAG}_kernel_stats_default_cmd.csv | cut -c1-160 ;;
    serial)
      S=$(slug "$ARG"); cd /tmp
      PF_STREAMS=1 timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pf_s -o bench -- python $R/bench.py $ARG --steps 4 --warmup 1 $NOLEGS --no-graphs > $R/gpurun_out/${TAG}_rocprof_serial.log 2>&1
      python $R/tools/prof_summary.py trace $(find /tmp/pf_s -name '*kernel_trace.csv' | head -1) $R/gpurun_out/${TAG}_kernels_serial${S:+_$S}.txt 10
      python $R/tools/prof_summary.py trace_steady $(find /tmp/pf_s -name '*kernel_trace.csv' | head -1) $R/gpurun_out/${TAG}_kernels_steady${S:+_$S}.txt 4 1
      head -n 4 $R/gpurun_out/${TAG}_kernels_steady${S:+_$S}.txt | cut -c1-400
      rm -rf /tmp/pf_s; head -n 16 $R/gpurun_out/${TAG}_kernels_serial${S:+_$S}.txt ;;
    pmc)
      S=$(slug "$ARG"); cd /tmp
      for C in FETCH_SIZE WRITE_SIZE; do
        PF_STREAMS=1 timeout 900 rocprofv3 --pmc $C --kernel-trace --output-format csv -d /tmp/pf_pmc_$C -o bench -- python $R/bench.py $ARG --steps 1 --warmup 0 $NOLEGS --no-graphs > $R/gpurun_out/${TAG}_pmc_$C.log 2>&1
        python $R/tools/prof_summary.py pmc $(find /tmp/pf_pmc_$C -name '*counter_collection.csv' | head -1) $R/gpurun_out/${TAG}${S:+_$S}_pmc_$C.txt
        rm -rf /tmp/pf_pmc_$C
        grep -E "k_conv_gemm|k_linear_ws|k_attention" $R/gpurun_out/${TAG}${S:+_$S}_pmc_$C.txt | cut -c1-200
      done
      python $R/tools/prof_summary.py traffic $R/gpurun_out/${TAG}${S:+_$S}_pmc_FETCH_SIZE.txt $R/gpurun_out/${TAG}${S:+_$S}_pmc_WRITE_SIZE.txt $R/gpurun_out/${TAG}${S:+_$S}_traffic.json
      cat $R/gpurun_out/${TAG}${S:+_$S}_traffic.json | cut -c1-400 ;;
    streams)
      cd /tmp
      timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/pf_e -o t -- python $R/bench.py --no-graphs --steps 3 --warmup 1 $NOLEGS > $R/gpurun_out/${TAG}_rocprof_eager.log 2>&1
      python $R/tools/trace_streams.py $(find /tmp/pf_e -name '*kernel_trace.csv' | head -1) $R/gpurun_out/${TAG}_streams.txt
      rm -rf /tmp/pf_e; tail -n 9 $R/gpurun_out/${TAG}_streams.txt ;;
    mfma)
      bash tools/gpu_mfma_instep.sh ${TAG} 2>&1 | tail -n 14 ;;
    gemm)
      python tools/gemm_bench.py --reps 20 ${ARG:+--shapes $ARG} 2>&1 | q | tee gpurun_out/${TAG}_gemm_microbench.txt | tail -n 40 ;;
    gemmphases)
      PF_GEMM8_PERSIST=0 python tools/gemm_bench.py --reps 5 --phases --shapes ${ARG:-conv64,lin320,ff1_320} 2>&1 | q | grep -v per-wave | tee gpurun_out/${TAG}_gemm_phases.txt ;;
    attn)
      python tools/attn_bench.py $ARG 2>&1 | q | tee gpurun_out/${TAG}_attn_microbench.txt ;;
    elem)
      python tools/elem_bench.py 2>&1 | q | tee gpurun_out/${TAG}_elem_microbench.txt ;;
    vae)
      python tools/vae_bench.py $ARG 2>&1 | q | tee gpurun_out/${TAG}_vae_bench.txt ;;
    train)
      timeout 400 python tools/train_bench.py --steps 3 2>&1 | q | tail -n 6 > gpurun_out/${TAG}_train_lora.txt; head -n 1 gpurun_out/${TAG}_train_lora.txt | cut -c1-300
      timeout 400 python tools/train_bench.py --layout-cond --steps 3 2>&1 | q | tail -n 6 > gpurun_out/${TAG}_train_layout_cond.txt; head -n 1 gpurun_out/${TAG}_train_layout_cond.txt | cut -c1-300 ;;
    sim)
      { for W in "2 0" "4 0,1" "8 0,1"; do read SW SR <<< "$W"; env $ARG python tools/sim_rank.py --world $SW --ranks $SR $SIM_ARGS 2>&1 | grep "^world"; done; } | tee -a gpurun_out/${TAG}_sim_ranks.txt ;;
    simprof)     # per-kernel table of ONE rank of a sharded layout (eager; 4 table-building + 1 warm-up + 3 timed passes)
      cd /tmp
      timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/pf_sp -o t -- python $R/tools/sim_rank.py ${ARG:---world 8 --ranks 1} --steps 3 --warmup 1 --no-graphs > $R/gpurun_out/${TAG}_simprof.log 2>&1
      python $R/tools/prof_summary.py trace $(find /tmp/pf_sp -name '*kernel_trace.csv' | head -1) $R/gpurun_out/${TAG}_simprof_$(slug "$ARG").txt 8
      rm -rf /tmp/pf_sp; head -n 24 $R/gpurun_out/${TAG}_simprof_$(slug "$ARG").txt ;;
    dist)
      NS="${ARG:-2 4 8}" bash tools/gpu_dist_dry.sh 2>&1 | tee gpurun_out/${TAG}_dist_dry.txt ;;
    *) echo "unknown section $NAME" ;;
  esac
done
