timeout 300 python -m pytest tests/test_prune2.py -m gpu -x -q 2>&1 | grep -v "^RCCL\|^HIP ver\|^ROCm\|^Hostname\|^Librccl" | tail -4
timeout 120 python tools/p2stats.py 1024 2>&1 | grep -v amdgpu | head -1
python bench.py --no-cpu --no-general --steps 200 2>/dev/null | python -c "
import json,sys
r=json.loads(sys.stdin.read()); print('default', r['ms_per_step'], r['ms_per_step_median'], r['frame_latency_ms'], r['roofline_timed_path']['kernel'], r['roofline_timed_path']['avg_launch_ms'])"
python tools/root32.py 2>/dev/null | grep "lanes 0 no_inv 0 *\(512\|octant\|256\) tiles     auto"
timeout 400 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "render3d" 2>&1 | grep -v "^RCCL\|^HIP ver\|^ROCm\|^Hostname\|^Librccl" | tail -3
