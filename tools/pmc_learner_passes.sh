cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/pmc_learner2; rm -rf $O; mkdir -p $O
timeout 500 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_learner_fetch -o pmc -- python $R/tools/pmc_learner_workload.py > $O/f.log 2>&1
timeout 500 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_learner_write -o pmc -- python $R/tools/pmc_learner_workload.py > $O/w.log 2>&1
PMC_TAIL=4 PMC_KEEP_TEMPLATE_ARGS=1 python $R/tools/pmc_summarise.py $O/pmc_learner_summary.json $O/pmc_learner_fetch $O/pmc_learner_write > $O/summary.log 2>&1
find $O -name "*counter_collection.csv" -delete; find $O -name "*agent_info.csv" -delete
tail -3 $O/f.log; tail -2 $O/summary.log
