#!/bin/bash
# Samples the shader clock and socket power (rocm-smi) twice a second while a command keeps the GPU busy:
#     tools/clock_watch.sh <seconds> <command ...>
n=$1
shift
"$@" > /dev/null 2>&1 &
pid=$!
for i in $(seq 1 $((2 * n))); do
    rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|Socket" | sed 's/.*(\([0-9]*Mhz\)).*/\1/; s/.*(W): //' | tr '\n' ' '
    echo
    sleep 0.5
done | sort | uniq -c | sort -k2 -n
kill $pid 2>/dev/null
wait $pid 2>/dev/null
