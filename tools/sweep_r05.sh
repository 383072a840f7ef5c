# GPU box, one call: the linked prune's tests and phase clocks, the bench line's numbers, the sizes of tools/root32.py, the render parity tests.
# (Round 5's A/B runs of this script - liveness sweep / queue / chain head start, posting / branch-free register scan, chunked / segmented tree
# scan, leaf kernel by layers / columns, every slab / the front slab, one / two root-level streams - are profiles/r05i/sweep*.txt; the switches
# they used are gone with the old forms.)      usage: gpurun -- 'bash tools/sweep_r05.sh'
F='^RCCL\|^HIP ver\|^ROCm\|^Hostname\|^Librccl\|amdgpu.ids'
timeout 300 python -m pytest tests/test_prune2.py -m gpu -x -q 2>&1 | grep -v "$F" | tail -3
timeout 120 python tools/p2stats.py 1024 2>&1 | grep -v "$F" | head -3 | cut -c1-200
FHIP_STATS=2 python tools/host_enqueue.py 1024 600 2>&1 | grep -v "$F" | tail -2 | cut -c1-330
python bench.py --no-cpu --steps 200 2>/dev/null | cut -c1-700
ROOT32_QUICK=1 python tools/root32.py 2>/dev/null | grep "lanes 0 no_inv 0 .* tiles     auto"
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_render_random.py tests/test_multi_gpu.py -m gpu -x -q -n 3 -k "render3d or random or octant or block or shard" 2>&1 | grep -v "$F" | tail -3
