#!/bin/bash
# round 4, very last session: the default bench line of the final build (drivers with page-locked arenas) and rocprofv3 kernel stats of
# the MSZIP launch shapes (parse waves with the literal ring)
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/r4h; mkdir -p $OUT; cd $R
export TMPDIR=/tmp
( timeout 600 python bench.py > $OUT/bench.json 2> $OUT/bench.err )
tail -c 400 $OUT/bench.json; echo
( cd /tmp && timeout 150 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/mz -o mz -- python $R/tools/bench_mszip_folder.py 4096 1 > $OUT/mz.log 2>&1 )
for f in $(find $OUT/mz -name "*kernel_stats.csv"); do cp $f $OUT/mszip_kernel_stats.csv; done
grep -v amdgpu $OUT/mz.log | tail -3; head -6 $OUT/mszip_kernel_stats.csv | cut -c1-40,150-260
