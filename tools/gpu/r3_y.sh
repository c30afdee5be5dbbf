#!/bin/bash
# what does the concurrent text tower cost the B = 36 forward?  (text features cached = no text tower at all: the upper bound)
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
for i in 1 2; do
timeout 300 python tools/step_probe.py --batch 36 --steps 20 --dtype fp16 2>&1 | grep "images/s"
timeout 300 python tools/step_probe.py --batch 36 --steps 20 --dtype fp16 --text-cache 2>&1 | grep "images/s"
done
