#!/bin/bash
# Run GPU steps one after another on a gpurun box, each under its own timeout, logging to gpurun_out/<tag>/<name>.log.
#   tools/gpu_steps.sh <tag> <name1> <seconds1> '<command1>' [<name2> <seconds2> '<command2>' ...]
# An ordinary failure (a red test) does not stop the following steps; a step that TIMES OUT or is KILLED does - no further GPU step is
# started after it (the box may be unhealthy).  The script's exit code is the worst of the steps'.
tag=$1; shift
out=gpurun_out/$tag
mkdir -p "$out"
worst=0
while [ $# -ge 3 ]; do
  name=$1; secs=$2; cmd=$3; shift 3
  echo "== $name (limit ${secs}s): $cmd"
  timeout -k 10 "$secs" bash -c "$cmd" > "$out/$name.log" 2>&1
  rc=$?
  echo "exit $rc" >> "$out/$name.log"
  echo "   -> exit $rc; tail:"; tail -n 6 "$out/$name.log" | cut -c1-400
  [ $rc -gt $worst ] && worst=$rc
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "== $name timed out / was killed: stopping here"; break; fi
done
exit $worst
