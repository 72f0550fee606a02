#!/usr/bin/env bash
# Copies what tools/round_check.sh <tag> left under gpurun_out/ into profiles/ under the round's name:
#   tools/install_profiles.sh <tag> <round, e.g. r05>
set -euo pipefail
TAG=$1; RND=$2
ROOT=$(cd "$(dirname "$0")/.." && pwd)
G=$ROOT/gpurun_out; P=$ROOT/profiles
for suf in "" "_C5"; do
    D=$G/profiles_$TAG$suf
    [[ -d $D ]] || continue
    cp $D/kernel_stats_$TAG$suf.csv $P/kernel_stats_$RND$suf.csv
    cp $D/pmc_sq_$TAG$suf.txt $P/pmc_sq_$RND$suf.txt
    cp $D/pmc_traffic_$TAG$suf.json $P/pmc_traffic_$RND$suf.json
    grep "^{\"metric\"" $D/bench_under_rocprof.log | tail -1 > $P/bench_under_rocprof_$RND$suf.json
done
for leg in default noflags nolazy nooverlap C5; do
    [[ -s $G/$TAG/bench_$leg.json ]] && cp $G/$TAG/bench_$leg.json $P/bench_${RND}_$leg.json
done
[[ -s $G/$TAG/step_timeline.txt ]] && cp $G/$TAG/step_timeline.txt $P/step_timeline_$RND.txt
ls -la $P | grep $RND
