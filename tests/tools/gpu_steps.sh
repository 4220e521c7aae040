#!/bin/bash
# developer tool: run GPU steps one after the other on the box; a step that is KILLED or TIMES OUT ends the script (no further
# GPU step is started after a hang), a step that merely fails (assertion) is recorded and the next one runs.
#   usage: source tests/tools/gpu_steps.sh; step <seconds> <logfile> <command...>
step() {
  local secs=$1 log=$2; shift 2
  echo "== $(date +%T) $* -> $log"
  mkdir -p "$(dirname "$log")"
  timeout -k 10 "$secs" "$@" > "$log" 2>&1
  local rc=$?
  echo "   rc=$rc"
  tail -n 3 "$log" | cut -c1-400 | sed 's/^/   | /'
  if [ $rc -ge 124 ]; then echo "step timed out or was killed: stopping"; exit $rc; fi
  return 0
}
