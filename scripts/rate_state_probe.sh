#!/bin/bash
# the Breakout step + render time series (scripts/power_probe.py) with rocm-smi's sensors sampled beside it
python scripts/power_probe.py > /tmp/probe.txt 2>&1 &
P=$!
sleep 6
for i in $(seq 1 12); do
  echo "--- sample $i $(date +%s.%N | cut -c1-14)"
  rocm-smi --showtemp --showclocks --showpower 2>/dev/null | grep -E "Temperature|clock level|Power" | sed 's/GPU\[0\]\s*: //' | tr '\n' ';'
  echo
  sleep 0.7
done
wait $P
cat /tmp/probe.txt
