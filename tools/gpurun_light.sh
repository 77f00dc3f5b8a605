#!/bin/bash
# kbench-only GPU call with a light snapshot: tests/golden (87 MB) left at home for THIS call only (push time is charged).
# usage: tools/gpurun_light.sh <timeout> '<command>'
cd "$(dirname "$0")/.."
printf 'tests/golden\n' > .gpurunignore
/usr/local/graft/bin/gpurun --timeout "$1" -- "$2"
rc=$?
rm -f .gpurunignore
exit $rc
