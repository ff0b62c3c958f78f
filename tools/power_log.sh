#!/bin/bash
# Samples rocm-smi power / clock / temperature while a command runs (on the MI355X box):
#   tools/power_log.sh <out.txt> <command...>
out=$1; shift
( while true; do rocm-smi --showpower --showclocks --showtemp --showperflevel 2>/dev/null | grep -E "Power|sclk|Temperature \(Sensor (edge|junction)|Performance" | tr -s ' ' | tr '\n' ';'; echo; sleep 0.5; done ) > $out &
SM=$!
"$@"
kill $SM 2>/dev/null
wait $SM 2>/dev/null
