#!/bin/bash
# Run a list of GPU steps on the gpurun box; each step under its own timeout, output to gpurun_out/.
# A step that fails normally (assertion, non-zero exit) does not stop the list; a step that TIMES OUT or is
# KILLED does (exit 124/137 or any signal): nothing further is started on a possibly wedged GPU.
# usage: tools/gpu_check.sh name1 'cmd1' secs1  name2 'cmd2' secs2 ...
mkdir -p gpurun_out
rc_all=0
while [ $# -ge 3 ]; do
  name=$1; cmd=$2; secs=$3; shift 3
  echo "=== [$name] $cmd (timeout ${secs}s)"
  start=$(date +%s)
  timeout -k 10 "$secs" bash -c "$cmd" > "gpurun_out/$name.log" 2>&1
  rc=$?
  echo "=== [$name] exit $rc after $(( $(date +%s) - start ))s"
  tail -n 12 "gpurun_out/$name.log"
  if [ $rc -ge 124 ]; then echo "=== [$name] timed out or was killed: stopping"; exit $rc; fi
  [ $rc -ne 0 ] && rc_all=$rc
done
exit $rc_all
