#!/bin/bash
# profiles/grun.sh <tag> <timeout-seconds> <command...>: one gpurun call; when no box / slot is free (exit code 3: nothing ran,
# nothing was charged) it is asked again after two minutes, up to ten times.  Any other outcome is final -- a command that ran
# is never run again from here.
tag=$1; lim=$2; shift 2
mkdir -p gpurun_out
for k in 1 2 3 4 5 6 7 8 9 10; do
  /usr/local/graft/bin/gpurun --timeout $lim -- "$@" > gpurun_out/${tag}_call.log 2>&1
  rc=$?
  [ $rc -ne 3 ] && break
  sleep 120
done
echo "grun $tag rc=$rc"; tail -40 gpurun_out/${tag}_call.log
exit $rc
