mkdir -p gpurun_out/r03t
bash tools/profile_round.sh r03 > gpurun_out/r03t/profile_round.log 2>&1
python bench.py > gpurun_out/r03t/bench.json 2> gpurun_out/r03t/bench.err; tail -c 300 gpurun_out/r03t/bench.json
python tools/config_times.py > gpurun_out/r03t/config_times.log 2>&1; tail -3 gpurun_out/r03t/config_times.log
bash tools/pmc_stalls.sh general > gpurun_out/r03t/pmc_stalls_general.log 2>&1; head -26 gpurun_out/pmc_stalls_general.txt
python tools/p2stats.py 2>&1 | grep "level 0"
