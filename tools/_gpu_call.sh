mkdir -p gpurun_out/r03g
bash tools/sweep_env.sh "FHIP_SLAB_LAYERS=2" "FHIP_SLAB_LAYERS=4" "FHIP_SLAB_LAYERS=8" "FHIP_SLAB_LAYERS=4 FHIP_SLAB_CONTEXTS=2" > gpurun_out/r03g/slab4.txt 2>&1
cat gpurun_out/r03g/slab4.txt
FHIP_SLAB_LAYERS=4 timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_render_random.py -m gpu -x -q > gpurun_out/r03g/tests4.log 2>&1; tail -3 gpurun_out/r03g/tests4.log
sed -i 's/for sl in (1, 2, 4)/for sl in (2, 4)/' tools/slab_probe.py; sed -i 's/"r03f"/"r03g"/g' tools/slab_probe.py
python tools/slab_probe.py 2>&1 | tail -9
