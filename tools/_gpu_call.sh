mkdir -p gpurun_out/r03e
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/r03e/tests.log 2>&1; echo "tests rc=$?" >> gpurun_out/r03e/tests.log
tail -6 gpurun_out/r03e/tests.log
bash tools/sweep_env.sh "FHIP_SLAB_LAYERS=1" "FHIP_SLAB_LAYERS=2" "FHIP_SLAB_LAYERS=4" "FHIP_SLAB_LAYERS=2 FHIP_SLAB_CONTEXTS=2" "FHIP_SLAB_LAYERS=4 FHIP_SLAB_CONTEXTS=2" > gpurun_out/r03e/slab_layers.txt 2>&1
cat gpurun_out/r03e/slab_layers.txt
timeout 600 python tools/mesh_probe.py > gpurun_out/r03e/mesh_probe.txt 2>&1; tail -5 gpurun_out/r03e/mesh_probe.txt
