#!/bin/bash
# ONE parametrised GPU-box script (replaces the per-call tools/gpu_rNN*.sh of earlier rounds):
#   gpurun --timeout T -- 'bash tools/gpu.sh <tag> <step> [<step> ...]'
# Everything is written under gpurun_out/<tag>/ (merged back by gpurun); copy what should be judged into profiles/<tag>/.
# Steps:
#   pytest[:<-k expr>]      the GPU suite (or the tests matching the expression)
#   smoke                   __graft_entry__.smoke()
#   bench[:<args>]          bench.py <args> -> bench[_<args>].json (+ .err)      (args with , for spaces, e.g. bench:--workload,train)
#   trace:<name>:<args>     rocprofv3 --kernel-trace --stats of bench.py <args> -> <name>_kernel_stats.csv (+ the bts:: lines on stdout)
#   steptrace:<name>:<marker>:<args>   rocprofv3 --kernel-trace of bench.py <args>; per-kernel table of the last 3 steps (delimited by the
#                           once-per-step kernel <marker>) -> <name>_steps.txt
#   ubench:<name>           tools/ubench/<name> -> <name>.txt
#   prof:<mode>[:K]         tools/profile.sh <tag> <mode> [K]   (PMC passes; summaries under gpurun_out/prof_<tag>/)
#   profstep:<workload>[:K[:layout]] tools/profile_step.sh <tag> <workload> [K] [nhwc|nchw]   (HBM traffic of every bts:: kernel of ONE fused training step)
#   py:<script>[:<args>]    python tools/<script>.py <args> -> <script>[_<args>].txt
set -u
cd "${GRAFT_REPO_ROOT:-.}"
TAG=$1; shift
O=gpurun_out/$TAG
mkdir -p $O
export TMPDIR=/tmp
for STEP in "$@"; do
  KIND=${STEP%%:*}; REST=${STEP#*:}; [ "$REST" = "$STEP" ] && REST=""
  echo "=== $STEP"
  case $KIND in
    pytest)
      if [ -n "$REST" ]; then timeout 1500 python -m pytest tests -m gpu -q --timeout 900 -rf --tb=short -k "${REST//,/ }" > $O/pytest_k.txt 2>&1; grep -E "^(E  |FAILED|ERROR|tests/.*(Error|assert))|passed|failed" $O/pytest_k.txt | cut -c1-240 | head -70
      else timeout 1700 python -m pytest tests -m gpu -q --timeout 900 --durations=8 -rf 2>&1 | tail -40 > $O/pytest_gpu.txt; tail -14 $O/pytest_gpu.txt; fi ;;
    smoke) timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 | tee $O/smoke.txt ;;
    bench)
      ARGS=${REST//,/ }; NAME=bench$(echo "$REST" | tr -c 'A-Za-z0-9\n' '_' | sed 's/__*/_/g;s/_$//')
      timeout 900 python bench.py $ARGS > $O/$NAME.json 2> $O/$NAME.err || tail -5 $O/$NAME.err
      python tools/bench_summary.py $O/$NAME.json ;;
    trace)
      NAME=${REST%%:*}; ARGS=${REST#*:}; ARGS=${ARGS//,/ }
      (cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/trace_$NAME -o trace -- python $GRAFT_REPO_ROOT/bench.py $ARGS > $GRAFT_REPO_ROOT/$O/trace_$NAME.log 2>&1)
      F=$(find $O/trace_$NAME -name "*kernel_stats.csv" | head -1)
      [ -n "$F" ] && cp $F $O/${NAME}_kernel_stats.csv && python tools/trace_table.py $O/${NAME}_kernel_stats.csv ${TRACE_STEPS:-1} 45
      rm -rf $O/trace_$NAME ;;
    steptrace)
      # steptrace:<name>:<marker kernel>:<bench args>   steady-state kernel table of the LAST 3 steps (tools/trace_steps.py)
      NAME=${REST%%:*}; R2=${REST#*:}; MARK=${R2%%:*}; ARGS=${R2#*:}; ARGS=${ARGS//,/ }
      (cd /tmp && timeout 1200 rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$O/strace_$NAME -o trace -- python $GRAFT_REPO_ROOT/bench.py $ARGS > $GRAFT_REPO_ROOT/$O/strace_$NAME.log 2>&1)
      F=$(find $O/strace_$NAME -name "*kernel_trace.csv" | head -1)
      [ -n "$F" ] && python tools/trace_steps.py $F $MARK 3 > $O/${NAME}_steps.txt 2>&1; head -60 $O/${NAME}_steps.txt
      rm -rf $O/strace_$NAME ;;
    ubench) timeout 600 tools/ubench/$REST > $O/$REST.txt 2>&1; cat $O/$REST.txt ;;
    prof) MODE=${REST%%:*}; KK=${REST#*:}; [ "$KK" = "$REST" ] && KK=""; timeout 1200 bash tools/profile.sh $TAG $MODE $KK 2>&1 | tail -20 ;;
    profstep) IFS=: read -r WL KK LAY <<< "$REST"; timeout 900 bash tools/profile_step.sh $TAG "$WL" "${KK:-}" "${LAY:-nhwc}" 2>&1 | tail -14 ;;
    py)
      NAME=${REST%%:*}; ARGS=${REST#*:}; [ "$ARGS" = "$REST" ] && ARGS=""
      timeout 900 python tools/$NAME.py ${ARGS//,/ } > $O/${NAME}$(echo "_$ARGS" | tr -c 'A-Za-z0-9\n' '_' | sed 's/__*/_/g;s/_$//').txt 2>&1; tail -40 $O/${NAME}*.txt ;;
    *) echo "unknown step $STEP" ;;
  esac
done
find $O -type f -size +8M -delete
