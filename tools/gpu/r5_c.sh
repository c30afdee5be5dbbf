#!/bin/bash
# lease C (round 5): correlation kernel v2 (2-row tiles, a whole tile of loads in flight per wave, T fragments pipelined), the re-based
# outlier / evaluator tests, kernel traces at B = 36 / 4 / 1
cd /root/repo; export TMPDIR=/tmp
O=gpurun_out/r5_c; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_ops.py -q -m gpu -k "corr" --timeout 300 > $O/tests_corr_op.log 2>&1; tail -3 $O/tests_corr_op.log
timeout 120 python tools/corr_bench.py > $O/corr_bench.txt 2>&1; cat $O/corr_bench.txt
rm -f gpurun_out/parity_table.txt
timeout 1200 python -m pytest tests/test_gpu_forward.py -q -m gpu -k "dedicated or outlier" --timeout 600 -s > $O/tests_fwd_parity.log 2>&1; tail -3 $O/tests_fwd_parity.log
cp gpurun_out/parity_table.txt $O/parity_table.txt 2>/dev/null
timeout 900 python -m pytest tests/test_gpu_evaluator.py -q -m gpu --timeout 800 -s > $O/tests_eval.log 2>&1; tail -3 $O/tests_eval.log
timeout 300 python bench.py --no-cpu-baseline --no-pmc-traffic --no-sweep --no-parity --dtype fp16 --steps 10 > $O/bench_fused.json 2> $O/bench_fused.err
cd /tmp
for b in 36 4 1; do
  timeout 300 rocprofv3 --kernel-trace --stats -d /root/repo/$O/prof_b$b -o t --output-format csv -- python /root/repo/bench.py --no-cpu-baseline --no-pmc-traffic --no-sweep --no-parity --dtype fp16 --batch $b --steps 4 --warmup 1 > /root/repo/$O/prof_b$b.log 2>&1
done
