#!/bin/bash
# gpurun, retried ONLY while no box / slot is free (exit code 3: nothing ran, nothing charged); any other outcome is final
# usage: scripts/gpurun_wait.sh TIMEOUT_S LOGFILE 'command'
T=$1; LOG=$2; shift 2
for i in $(seq 1 40); do
    /usr/local/graft/bin/gpurun --timeout "$T" -- "$@" > "$LOG" 2>&1
    rc=$?
    [ $rc -ne 3 ] && break
    sleep 120
done
echo "gpurun exit $rc" >> "$LOG"
exit $rc
