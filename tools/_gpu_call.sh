mkdir -p gpurun_out/r03z
R=$PWD
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/t1; rocprofv3 --kernel-trace --output-format csv -d /tmp/t1 -o t -- python $R/tools/one_frame.py > $R/gpurun_out/r03z/one_frame_run.log 2>&1
cd $R; python tools/timeline.py /tmp/t1 1 1 > gpurun_out/r03z/timeline_one_frame.txt 2>&1
cd /tmp; rm -rf /tmp/t2; rocprofv3 --kernel-trace --output-format csv -d /tmp/t2 -o t -- python $R/tools/one_frame.py general > $R/gpurun_out/r03z/one_frame_general_run.log 2>&1
cd $R; python tools/timeline.py /tmp/t2 1 1 > gpurun_out/r03z/timeline_one_frame_general.txt 2>&1
head -3 gpurun_out/r03z/timeline_one_frame.txt; tail -16 gpurun_out/r03z/timeline_one_frame.txt; head -1 gpurun_out/r03z/timeline_one_frame_general.txt
