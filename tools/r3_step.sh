# usage (inside a gpurun command): source tools/r3_step.sh; step <seconds> <logfile> <command...>
# runs one GPU step under its own timeout; a step that was killed at its limit (or died on a signal) stops the whole call:
# no further GPU work is started after a hang.
step() {
  local limit=$1 log=$2; shift 2
  echo "== $(date +%T) step: $* (limit ${limit}s) -> $log"
  timeout -k 10 "$limit" "$@" > "$log" 2>&1
  local rc=$?
  echo "== $(date +%T) rc=$rc"
  if [ $rc -eq 124 ] || [ $rc -ge 128 ]; then echo "step killed (rc=$rc): stopping"; exit $rc; fi
  return $rc
}
