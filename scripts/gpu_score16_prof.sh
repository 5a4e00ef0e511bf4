export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$PWD}
rm -rf $R/gpurun_out/sprof; mkdir -p $R/gpurun_out/sprof
cd /tmp
NQ=16 REPS=30 timeout 300 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/sprof -o trace -- python $R/scripts/score_bench.py > $R/gpurun_out/sprof.log 2>&1
cd $R
DB=$(find gpurun_out/sprof -name '*.db' | head -1)
python scripts/prof_summary.py $DB 14 | cut -c1-200 | tee gpurun_out/score16_prof_summary.csv
python scripts/prof_gaps.py $DB 400 | tee -a gpurun_out/score16_prof_summary.csv
rm -rf gpurun_out/sprof
