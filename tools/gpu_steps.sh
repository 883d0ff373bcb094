#!/bin/bash
# Runs named GPU steps one after the other on a gpurun box: each under its own timeout, output to
# gpurun_out/<name>.txt; stops at the first step that timed out or was killed (never starts another
# GPU step after a hang), carries on after an ordinary non-zero exit (a failing assertion).
#   usage: bash tools/gpu_steps.sh <seconds per step> name1 "command 1" name2 "command 2" ...
T=$1; shift
mkdir -p gpurun_out
while [ $# -ge 2 ]; do
  name=$1; cmd=$2; shift 2
  start=$(date +%s)
  timeout -k 10 "$T" bash -c "$cmd" > "gpurun_out/$name.txt" 2> "gpurun_out/$name.err"
  rc=$?
  echo "step $name: rc=$rc, $(( $(date +%s) - start )) s"
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "step $name timed out or was killed: stopping"; exit $rc; fi
done
exit 0
