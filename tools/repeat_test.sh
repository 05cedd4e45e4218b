#!/bin/bash
# tools/repeat_test.sh <tag> <times> <seconds each> <pytest node>: a test several times in fresh processes, full log kept for every red run (gpurun_out/<tag>/)
tag=$1; n=$2; secs=$3; node=$4
mkdir -p gpurun_out/$tag
for i in $(seq 1 $n); do
  timeout -k 10 $secs python -m pytest "$node" -x -q > gpurun_out/$tag/run$i.log 2>&1
  rc=$?
  echo "run $i: exit $rc: $(grep -v amdgpu gpurun_out/$tag/run$i.log | tail -1)"
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "timed out: stopping"; break; fi
  [ $rc -eq 0 ] && rm gpurun_out/$tag/run$i.log
done
exit 0
