cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout -k 10 400 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d /tmp/pmc_try -o p -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu > $R/gpurun_out/pmc_try.log 2>&1 < /dev/null
echo "exit $?" >> $R/gpurun_out/pmc_try.log
tail -n 40 $R/gpurun_out/pmc_try.log
