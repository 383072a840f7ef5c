mkdir -p gpurun_out/r03r
R=$PWD
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/tg; rocprofv3 --kernel-trace --output-format csv -d /tmp/tg -o t -- python $R/bench.py --steps 40 --warmup 3 --no-cpu --no-general > $R/gpurun_out/r03r/trace_run.log 2>&1
cd $R
python tools/timeline.py /tmp/tg 3 +20 > gpurun_out/r03r/timeline_queued_default.txt 2>&1
tail -8 gpurun_out/r03r/timeline_queued_default.txt; head -3 gpurun_out/r03r/timeline_queued_default.txt
rm -rf /tmp/tg; cd /tmp; rocprofv3 --kernel-trace --output-format csv -d /tmp/tg -o t -- python $R/bench.py --steps 40 --warmup 3 --no-cpu --only-general > $R/gpurun_out/r03r/trace_run_general.log 2>&1
cd $R
python tools/timeline.py /tmp/tg 3 +20 > gpurun_out/r03r/timeline_queued_general.txt 2>&1
tail -6 gpurun_out/r03r/timeline_queued_general.txt; head -3 gpurun_out/r03r/timeline_queued_general.txt
