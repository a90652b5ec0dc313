#!/bin/bash
# A/B of the read-spectrum paths on the config-4 read set + kernel trace of the default path: bash tools/sketch_ab.sh <out_dir>
set -e
out=${1:-gpurun_out/sketch_ab}; mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
python3 tools/score_profile.py 0 generic > $out/generic.txt 2>&1
python3 tools/score_profile.py 0 exact > $out/exact.txt 2>&1
python3 tools/score_profile.py 0 > $out/buckets.txt 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof -o sk -- python3 tools/score_profile.py 0 > $out/prof.log 2>&1
f=$(find $out/prof -name '*kernel_stats.csv' | head -1); cp "$f" $out/kernel_stats.csv
cat $out/generic.txt $out/exact.txt $out/buckets.txt
