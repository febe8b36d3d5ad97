#!/bin/bash
# round 5: the profile sets committed under profiles/r5/ (headline, cfg2, cfg4, top100)
cd "$(dirname "$0")/.."
bash tools/profile_round.sh r5_headline > gpurun_out/prof_r5_headline.log 2>&1; tail -2 gpurun_out/prof_r5_headline.log
bash tools/profile_round.sh r5_cfg2 --config cfg2 > gpurun_out/prof_r5_cfg2.log 2>&1; tail -2 gpurun_out/prof_r5_cfg2.log
bash tools/profile_round.sh r5_top100 --config top100 > gpurun_out/prof_r5_top100.log 2>&1; tail -2 gpurun_out/prof_r5_top100.log
bash tools/profile_round.sh r5_cfg4 --config cfg4 > gpurun_out/prof_r5_cfg4.log 2>&1; tail -2 gpurun_out/prof_r5_cfg4.log
find gpurun_out/prof_r5_* -name "*.csv" -size +8M -delete
