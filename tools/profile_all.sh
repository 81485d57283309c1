#!/bin/bash
# ONE script for the profiles of the shipped build.  Runs on the GPU box (via gpurun): for each of the five bench workloads, rocprofv3
# --kernel-trace --stats of the bench command, then the counter passes of the same command (each its own run, kernel-trace only: FETCH_SIZE and WRITE_SIZE cannot share
# a pass on gfx950, and gpurun refuses --pmc combined with the other trace domains) -> gpurun_out/prof_*.  Back in the build container,
# `python tools/summarize_all.py rNN` copies the summaries into profiles/rNN_* and rewrites profiles/traffic.json; tests/test_profiles.py then checks every
# kernel name in profiles/rNN_* against the instantiations of HEAD's libearl_hip.so.   usage: bash tools/profile_all.sh [workload ...]   (default: all five)
set -u
export TMPDIR=/tmp
OUT=$PWD/gpurun_out
mkdir -p $OUT
WORKLOADS=${@:-tabletop sawyer_door sawyer_peg kitchen minitaur}
SQ1="SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY"
SQ2="SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM"
SQ3="SQ_THREAD_CYCLES_VALU SQ_INSTS_VALU SQ_BUSY_CYCLES SQ_WAVES"
SQ4="SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_SALU"
prof() {   # prof <dir tag> <bench args...> ; counter sets in PASSES (newline separated)
  local tag=$1; shift
  rm -rf $OUT/prof_${tag}_stats $OUT/prof_${tag}_pmc*
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_${tag}_stats -- python3 bench.py "$@" > $OUT/prof_${tag}_stats.log 2>&1
  echo "$tag stats rc=$?"
  local i=0
  while IFS= read -r C; do
    [ -z "$C" ] && continue
    i=$((i+1))
    rocprofv3 --kernel-trace --pmc $C --output-format csv -d $OUT/prof_${tag}_pmc$i -- python3 bench.py "$@" > $OUT/prof_${tag}_pmc$i.log 2>&1
    echo "$tag pmc pass $i ($C) rc=$?"
  done <<< "$PASSES"
}
for W in $WORKLOADS; do
  case $W in
    tabletop)     # the default line's own launches only (25 launches of 28 evaluation episodes each, own actions): the HBM-bound kernel -> traffic
      PASSES=$'FETCH_SIZE\nWRITE_SIZE' prof tabletop --no-cpu --no-step-api --no-sawyer --no-kitchen --no-minitaur --no-single
      rm -rf $OUT/prof_stats $OUT/prof_fetch $OUT/prof_write     # (the names tools/summarize_profiles.py reads)
      mv $OUT/prof_tabletop_stats $OUT/prof_stats; mv $OUT/prof_tabletop_pmc1 $OUT/prof_fetch; mv $OUT/prof_tabletop_pmc2 $OUT/prof_write ;;
    sawyer_door|sawyer_peg)
      PASSES="$SQ1"$'\n'"$SQ2"$'\n'"$SQ3"$'\n'"$SQ4"$'\nFETCH_SIZE\nWRITE_SIZE' prof $W --workload $W --steps 3 --warmup 1 --no-cpu ;;
    kitchen)
      PASSES="$SQ1"$'\n'"$SQ2"$'\n'"$SQ3"$'\nFETCH_SIZE\nWRITE_SIZE' prof kitchen --workload kitchen --steps 1 --warmup 1 --no-cpu --no-step-api ;;
    minitaur)
      PASSES="$SQ1"$'\n'"$SQ2"$'\n'"$SQ3"$'\nFETCH_SIZE\nWRITE_SIZE' prof minitaur --workload minitaur --steps 1 --warmup 1 --no-cpu ;;
  esac
done
# keep what travels back small: the per-dispatch traces are not needed, the stats and counter CSVs are
find $OUT -path '*prof_*' \( -name '*_kernel_trace.csv' -o -name '*_agent_info.csv' \) -size +4M -delete 2>/dev/null
du -sh $OUT/prof_* 2>/dev/null | tail -40
