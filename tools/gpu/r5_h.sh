#!/bin/bash
# lease H (round 5): what the deterministic-reduction mode costs a training step (tools/train_bench.py, B = 8); the bench line with the
# batch-72 sweep point
cd /root/repo; export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
O=gpurun_out/r5_h; mkdir -p $O
for d in 0 1 0 1; do
  echo "== LSEG_DETERMINISTIC=$d" >> $O/train_det.txt
  LSEG_DETERMINISTIC=$d timeout 200 python tools/train_bench.py --steps 20 2>&1 | tail -1 >> $O/train_det.txt
done
cat $O/train_det.txt
( timeout 900 python bench.py 2>&1 | grep "^{" ) > $O/bench_line.json; cut -c1-200 $O/bench_line.json
python - <<'PY'
import json
d = json.loads(open("/root/repo/gpurun_out/r5_h/bench_line.json").read().strip().splitlines()[-1])
print(d["value"], d["batch_sweep_images_per_sec"], d.get("extra_legs"))
PY
