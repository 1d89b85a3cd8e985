# GPU clock / power while the pipelined b = 1 step runs (is the aggregate MFMA rate clock-limited?)
PIPE_DEPTH=4 PIPE_STEPS=1500 python tools/pipeline_probe.py > /tmp/pp.log 2>&1 &
PP=$!
sleep 14
for i in 1 2 3 4; do rocm-smi --showclocks --showpower 2>/dev/null | grep -i "sclk\|mclk\|power\|fclk" | head -6; echo --; sleep 2; done
wait $PP
grep "in flight\|back to back" /tmp/pp.log
