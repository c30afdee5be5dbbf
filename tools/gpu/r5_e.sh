#!/bin/bash
# lease E (round 5): the one-stack evaluator (tests + the bench leg), the outlier fixture rows with the re-based bars
cd /root/repo; export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
O=gpurun_out/r5_e; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_evaluator.py -q -m gpu --timeout 800 -s > $O/tests_eval.log 2>&1; tail -4 $O/tests_eval.log
timeout 900 python -m pytest tests/test_gpu_forward.py -q -m gpu -k "outlier" --timeout 600 -s > $O/tests_outlier.log 2>&1; tail -3 $O/tests_outlier.log; grep "relative rms" $O/tests_outlier.log
timeout 600 python - > $O/eval_leg.json 2> $O/eval_leg.err <<'PY'
import json, os, sys, torch
sys.argv = ["bench.py"]
sys.path.insert(0, "/root/repo")
import bench
from lseg_hip.config import get_config
from lseg_hip.synth import synthetic_state_dict, read_labels
args = bench.parse()
cfg = get_config(args.backbone)
sd = synthetic_state_dict(cfg, seed=0)
labels = read_labels(os.path.join(bench.ROOT, "lang-seg_amd", "label_files", "ade20k_objectInfo150.txt"))[:150]
print(json.dumps(bench.eval_leg(args, cfg, sd, labels, "fp16", torch.cuda.synchronize)))
PY
cat $O/eval_leg.json | cut -c1-700
