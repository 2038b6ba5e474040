#!/bin/bash
# round 3, fifth GPU session: the whole suite with the captured prints, then the profile of the round
OUT=gpurun_out/${1:-r03e}
mkdir -p $OUT
export OMP_NUM_THREADS=1
timeout -k 10 900 python -m pytest tests -m gpu -q -rP -p no:cacheprovider --durations=10 > $OUT/gputests.log 2>&1
rc=$?
echo "pytest rc=$rc" | tee -a $OUT/gputests.log
grep -E "passed|failed|worst five|strict mode|1024\^2:|config 3:|launches per solve|true residuals|which [01]|iterations direct|FAILED" $OUT/gputests.log | tail -30
if [ $rc -gt 1 ]; then exit $rc; fi
timeout -k 10 1000 bash tools/profile_round.sh ${1:-r03e}
