#!/bin/bash
# Run GPU steps one after another; stop at the first one that times out or is killed (never start a GPU step behind a hung one).
# usage: tools/gpu_steps.sh "<seconds> <logfile> <command...>" ...
for spec in "$@"; do
    set -- $spec
    secs=$1; log=$2; shift 2
    echo "== $* (limit ${secs}s) -> $log"
    timeout -k 10 "$secs" "$@" > "$log" 2>&1
    rc=$?
    echo "   exit $rc"; tail -n 3 "$log" | sed 's/^/   | /'
    if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "step timed out / was killed: stopping"; exit $rc; fi
done
exit 0
