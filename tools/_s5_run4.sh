# session 5 / run 4: the dispatch ramp of a k_step-shaped launch (stand-alone probe) and the launch's own ramp inside the loop
mkdir -p gpurun_out/s5
O=gpurun_out/s5/run4.txt; : > $O
echo "== dispatch_ramp_probe" >> $O
timeout 120 tools/native/dispatch_ramp_probe >> $O 2>&1
echo "== timeline, default order" >> $O
timeout 300 python tools/step_timeline.py 2>&1 | tail -33 | head -8 >> $O
echo "== timeline, CATAN_STEP_BIN_ORDER=1" >> $O
CATAN_STEP_BIN_ORDER=1 timeout 300 python tools/step_timeline.py 2>&1 | tail -33 | head -8 >> $O
cat $O
