#!/bin/bash
# exploratory: sample rocm-smi (power, clocks, temperature) while a command runs:  tools/power_probe.sh <outfile> <command...>
out=$1; shift
"$@" > ${out}.cmd.log 2>&1 &
pid=$!
while kill -0 $pid 2>/dev/null; do
  /opt/rocm/bin/rocm-smi --showpower --showclocks --showtemp --json 2>/dev/null | head -c 1500 >> $out; echo >> $out
  sleep 0.5
done
wait $pid
