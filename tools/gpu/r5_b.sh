#!/bin/bash
# lease B (round 5): the dedicated correlation kernel (op test, engine A/B, standalone timing, bench with / without it), the outlier-weights
# parity rows, the crop-480 evaluator test, the new bench legs.
cd /root/repo; export TMPDIR=/tmp
O=gpurun_out/r5_b; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_ops.py -q -m gpu -k "corr" --timeout 300 > $O/tests_corr_op.log 2>&1; tail -3 $O/tests_corr_op.log
timeout 120 python tools/corr_bench.py > $O/corr_bench.txt 2>&1; cat $O/corr_bench.txt
rm -f gpurun_out/parity_table.txt
timeout 1200 python -m pytest tests/test_gpu_forward.py -q -m gpu -k "dedicated or baseline_configs or masks_match" --timeout 600 -s > $O/tests_fwd_parity.log 2>&1; tail -3 $O/tests_fwd_parity.log
cp gpurun_out/parity_table.txt $O/parity_table.txt 2>/dev/null
timeout 600 python -m pytest tests/test_gpu_evaluator.py tests/test_zz_gpu_random_net_gradients.py "tests/test_gpu_train.py::test_atomic_sums_stay_within_rounding_of_the_deterministic_ones" -q -m gpu --timeout 500 -s > $O/tests_misc.log 2>&1; tail -3 $O/tests_misc.log
LSEG_CORR_GENERIC=1 timeout 300 python bench.py --no-cpu-baseline --no-pmc-traffic --no-sweep --dtype fp16 --steps 10 > $O/bench_generic.json 2> $O/bench_generic.err
timeout 300 python bench.py --no-cpu-baseline --no-pmc-traffic --no-sweep --dtype fp16 --steps 10 > $O/bench_fused.json 2> $O/bench_fused.err
timeout 900 python bench.py > $O/bench_full.json 2> $O/bench_full.err; tail -c 600 $O/bench_full.err
cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d /root/repo/$O/prof -o corr --output-format csv -- python /root/repo/bench.py --no-cpu-baseline --no-pmc-traffic --no-sweep --no-parity --dtype fp16 --steps 3 --warmup 1 > /root/repo/$O/prof.log 2>&1
cd /root/repo; ls $O/prof 2>/dev/null | head
