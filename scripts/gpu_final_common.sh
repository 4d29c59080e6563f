# sourced by the round-4 evidence scripts: run one GPU step under its own limit; a step that was killed at its limit ends the call
step() {   # limit, label, command...
  local limit=$1 label=$2; shift 2
  timeout -k 10 "$limit" "$@"
  local rc=$?
  echo "$label rc=$rc"
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "$label was killed at its limit: no further GPU step in this call"; exit $rc; fi
  return $rc
}
