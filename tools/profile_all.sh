#!/bin/bash
# rocprofv3 evidence for every bench configuration of one round (run on the GPU box from the repo root):
#   tools/profile_all.sh r03   ->  gpurun_out/prof_${R}_{noblank,noblank2048,binary,blank} + summaries printed
set -u
R=${1:-r03}
O=gpurun_out
mkdir -p $O
bash tools/profile.sh $O/prof_${R}_noblank                                                   > $O/prof_${R}_noblank.log 2>&1
echo "noblank done"
bash tools/profile.sh $O/prof_${R}_noblank2048 --scaling strong --global-batch 2048          > $O/prof_${R}_noblank2048.log 2>&1
echo "noblank2048 done"
bash tools/profile.sh $O/prof_${R}_binary --variant binary                                   > $O/prof_${R}_binary.log 2>&1
echo "binary done"
bash tools/profile.sh $O/prof_${R}_blank --variant blank                                     > $O/prof_${R}_blank.log 2>&1
echo "blank done"
python tools/summarize_prof.py $O/prof_${R}_noblank     $O/sum_${R}/${R}_noblank_cfg2   noblank_B256  24268800  r16_kernel   > /dev/null
python tools/summarize_prof.py $O/prof_${R}_noblank2048 $O/sum_${R}/${R}_noblank_B2048  noblank_B2048 194150400 r16_kernel   > /dev/null
python tools/summarize_prof.py $O/prof_${R}_binary      $O/sum_${R}/${R}_binary_cfg3    binary_B256   24268800  binary_      > /dev/null
python tools/summarize_prof.py $O/prof_${R}_blank       $O/sum_${R}/${R}_blank_cfg5     blank_B64     512000000 blank_       > /dev/null
for v in noblank noblank2048 binary blank; do cp $(find $O/prof_${R}_$v/trace -name "*kernel_stats.csv" | head -1) $O/sum_${R}/${R}_${v}_kernel_stats.csv; done
# the raw per-dispatch CSVs are large: keep the summaries, the stats and the bench lines
for v in noblank noblank2048 binary blank; do cp $O/prof_${R}_$v/bench_trace.json $O/sum_${R}/${v}_under_rocprof.json; rm -rf $O/prof_${R}_$v/pmc_* $O/prof_${R}_$v/trace; done
cat $O/sum_${R}/traffic.json
