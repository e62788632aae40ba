#!/bin/bash
# Samples clocks / power of GPU 0 (rocm-smi / amd-smi, whichever answers as an ordinary user) while a bench leg runs.
#   bash tools/power_probe.sh <out> -- <command...>
OUT=$1; shift; shift
"$@" > $OUT.cmd.log 2>&1 &
PID=$!
sleep ${POWER_PROBE_DELAY:-25}
for i in $(seq 1 ${POWER_PROBE_N:-12}); do
  kill -0 $PID 2>/dev/null || break
  echo "--- sample $i" >> $OUT
  timeout 10 rocm-smi -d 0 --showclocks --showpower --showtemp --showperflevel 2>&1 | grep -v "^=\|^$" >> $OUT
  sleep 1
done
wait $PID
echo "--- idle" >> $OUT
sleep 3
timeout 10 rocm-smi -d 0 --showclocks --showpower --showtemp 2>&1 | grep -v "^=\|^$" >> $OUT
tail -1 $OUT.cmd.log | cut -c1-300 >> $OUT
