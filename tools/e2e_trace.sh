cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
python $R/tools/e2e_trace.py 3 60
python $R/tools/e2e_trace.py 1 40
rm -rf /tmp/e2etr; rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d /tmp/e2etr -o e2e -- python $R/tools/e2e_trace.py 3 60 2>&1 | tail -2
E2E_SUMMARISE=/tmp/e2etr python $R/tools/e2e_trace.py
head -3 $(find /tmp/e2etr -name "*memory_copy_trace.csv" | head -1)
rm -rf /tmp/e2etr
