#!/bin/bash
# tools/profile_r05_first.sh -- round 5's first GPU call: the whole GPU suite with its durations, the bench line, and the SQ counters of
# the headline kernel before (round 4's library, build/exp/libfpx_r04_product.so) and after the ISA fixes
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r05a
rm -rf $O; mkdir -p $O
cd $R
timeout 1500 python -m pytest tests -q -m gpu --durations=60 -p no:cacheprovider > $O/pytest.log 2>&1
echo "pytest rc $?" >> $O/pytest.log
python bench.py > $O/bench.json 2> $O/bench.err; tail -c 3000 $O/bench.err > $O/bench.tail; rm -f $O/bench.err
mkdir -p $O/sq_after $O/sq_before
bash tools/pmc_sq.sh r05a/sq_after > $O/sq_after.txt 2>&1
FPX_LIB=$R/acoustid-index_amd/build/exp/libfpx_r04_product.so bash tools/pmc_sq.sh r05a/sq_before > $O/sq_before.txt 2>&1
find $O -name "*.csv" -size +2M -delete
du -sh $O
