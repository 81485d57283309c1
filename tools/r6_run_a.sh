mkdir -p gpurun_out/r6a
python -m pytest tests -m gpu -x -q > gpurun_out/r6a/gputests.log 2>&1; echo "pytest rc $?" >> gpurun_out/r6a/gputests.log
python bench.py > gpurun_out/r6a/bench_line.json 2> gpurun_out/r6a/bench_err.log; cp bench_full.json gpurun_out/r6a/
timeout 600 python tools/strict_bound.py > gpurun_out/r6a/strict_bound.json 2>&1
(timeout 300 python tools/bench_mt_variant.py ship 4096 200; timeout 300 python tools/bench_mt_variant.py b2 4096 200) > gpurun_out/r6a/mt_variants.txt 2>&1
tail -3 gpurun_out/r6a/gputests.log; wc -c gpurun_out/r6a/bench_line.json; cat gpurun_out/r6a/mt_variants.txt; tail -70 gpurun_out/r6a/strict_bound.json
timeout 900 python tools/contraction_cost.py > gpurun_out/r6a/contraction_cost.json 2>&1
tail -50 gpurun_out/r6a/contraction_cost.json
