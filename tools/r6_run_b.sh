mkdir -p gpurun_out/r6b
python -m pytest tests -m gpu -x -q > gpurun_out/r6b/gputests.log 2>&1; echo "pytest rc $?" >> gpurun_out/r6b/gputests.log
timeout 600 python tools/strict_bound.py > gpurun_out/r6b/strict_bound.json 2>&1
(timeout 300 python tools/bench_mt_variant.py ship 4096 200; timeout 300 python tools/bench_mt_variant.py b2 4096 200; timeout 300 python tools/bench_mt_variant.py b2 8192 200;  timeout 300 python tools/bench_mt_variant.py ship 8192 200) > gpurun_out/r6b/mt_variants.txt 2>&1
tail -3 gpurun_out/r6b/gputests.log; cat gpurun_out/r6b/mt_variants.txt; tail -80 gpurun_out/r6b/strict_bound.json
