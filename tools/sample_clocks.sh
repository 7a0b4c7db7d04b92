#!/bin/bash
# Samples the GPU's shader clock and socket power (rocm-smi) twice a second while a command runs: is the path power-limited?
# usage: sample_clocks.sh out.txt -- command ...
OUT=$1; shift; shift
( while true; do /opt/rocm/bin/rocm-smi --showclocks --showpower --showtemp 2>/dev/null | grep -E "sclk|mclk|fclk|Power|Temperature \(Sensor (junction|hotspot)" | tr '\n' ' ' ; echo; sleep 0.5; done ) > $OUT 2>&1 &
SP=$!
"$@"
RC=$?
kill $SP 2>/dev/null
exit $RC
