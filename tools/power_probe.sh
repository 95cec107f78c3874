#!/bin/bash
# Board power and clocks while viterbi_ck runs back to back (40 000 pairs per launch, ~12 s): is the kernel's clock
# (1.8-1.9 GHz) the power limit?  usage (GPU box, repo root): bash tools/power_probe.sh
cd "$(dirname "$0")/.."
python3 tools/fill_loop.py 40000 800 > /tmp/fill_loop.out 2>&1 &
PID=$!
sleep 4
for i in 1 2 3 4 5; do
    rocm-smi --showpower --showclocks --showperflevel --showmaxpower 2>/dev/null | grep -E "Power|sclk|mclk|fclk|Max Graphics|Performance" | sed 's/^/  /' | tr '\n' ';'
    echo
    sleep 1
done
wait $PID
cat /tmp/fill_loop.out
echo "idle:"; sleep 2
rocm-smi --showpower --showclocks --showmaxpower 2>/dev/null | grep -E "Power|sclk" | sed 's/^/  /' | tr '\n' ';'; echo
