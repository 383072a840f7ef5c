mkdir -p gpurun_out/r03z
python bench.py > gpurun_out/r03z/bench2.json 2> gpurun_out/r03z/bench2.err; tail -c 100 gpurun_out/r03z/bench2.json
