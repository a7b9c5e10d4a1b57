#!/bin/bash
# Wrapper around gpurun that keeps a record of every GPU-box command of the round (profiles/r04_gpu_calls.log):
# a fault can then be tied to the exact command and tree state that produced it.
#   tools/gpu.sh [--timeout S] '<command>'
TO=600
if [ "$1" = "--timeout" ]; then TO=$2; shift 2; fi
LOG="$(dirname "$0")/../profiles/r04_gpu_calls.log"
echo "$(date -u +%FT%TZ) head=$(git -C "$(dirname "$0")/.." rev-parse --short HEAD) dirty=$(git -C "$(dirname "$0")/.." status --porcelain | wc -l) timeout=$TO :: $*" >> "$LOG"
/usr/local/graft/bin/gpurun --timeout "$TO" -- "$@"
rc=$?
echo "$(date -u +%FT%TZ)   -> exit $rc" >> "$LOG"
exit $rc
