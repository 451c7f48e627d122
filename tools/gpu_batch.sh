#!/bin/bash
# Runs the steps of a GPU session one after the other (one `gpurun` call = one box): each line of the step file is
# `<seconds> <log name> <command...>`; a step that fails is recorded and the next one runs, a step that TIMES OUT or is
# killed ends the session (nothing else is started on a GPU that may be hung).  Logs land in gpurun_out/.
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
while IFS= read -r line; do
    [ -z "$line" ] && continue
    case "$line" in \#*) continue;; esac
    secs=${line%% *}; rest=${line#* }; name=${rest%% *}; cmd=${rest#* }
    echo "== [$name] $cmd"
    start=$(date +%s)
    timeout -k 10 "$secs" bash -c "$cmd" > "gpurun_out/$name" 2>&1
    rc=$?
    echo "== [$name] rc=$rc in $(( $(date +%s) - start )) s"
    tail -n 3 "gpurun_out/$name"
    if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "== [$name] timed out: session ends here"; exit 1; fi
done < "$1"
exit 0
