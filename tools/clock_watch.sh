#!/bin/bash
# Runs on the GPU box: sample sclk / power with rocm-smi while a command runs.  Usage: clock_watch.sh <label> <cmd...>
label=$1; shift
( for i in $(seq 1 400); do rocm-smi --showclocks --showpower --json 2>/dev/null | python3 -c "
import sys, json
try:
    j = json.load(sys.stdin)['card0']
    print('$label', {k: v for k, v in j.items() if 'sclk' in k.lower() or 'ower' in k})
except Exception as e:
    print('smi parse', e)
"; sleep 0.25; done ) > /tmp/clk_$label.txt 2>&1 &
W=$!
"$@"
kill $W 2>/dev/null
sort /tmp/clk_$label.txt | uniq -c | sort -rn | head -8
