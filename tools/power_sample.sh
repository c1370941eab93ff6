#!/bin/bash
# tools/power_sample.sh <out.jsonl> -- <command ...>: run the command while sampling the socket power, the shader / memory clocks and
# the power cap (rocm-smi, as fast as it answers: a few samples per second) -- to tell a power-limited kernel (clock pulled down to
# the cap) from an issue-limited one.  One JSON object per sample: {"t": seconds since start, "smi": {...}}.
OUT=$1; shift; shift
"$@" &
PID=$!
T0=$(date +%s.%N)
: > $OUT
/opt/rocm/bin/rocm-smi -M --json 2>/dev/null | head -c 2000 >> $OUT; echo >> $OUT
while kill -0 $PID 2>/dev/null; do
  T=$(echo "$(date +%s.%N) - $T0" | bc)
  S=$(/opt/rocm/bin/rocm-smi -P -g --json 2>/dev/null | tr -d '\n')
  echo "{\"t\": $T, \"smi\": $S}" >> $OUT
done
wait $PID
