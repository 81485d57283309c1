export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
for T in 8 16 40 200; do
  rm -rf gpurun_out/fx_$T
  TUNE_T=$T rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/fx_$T -- python3 tools/tune_rollout.py 0 4096 > /dev/null 2>&1
  f=$(find gpurun_out/fx_$T -name '*kernel_stats.csv' | head -1)
  echo "T=$T $(grep rollout_ws $f | cut -d, -f2-8)"
done
