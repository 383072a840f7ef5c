# GPU box, one call: the changes of round 5's last third, each against its old form (FHIP_DEBUG_BITS: 4 liveness sweep, 32 work queue without the chain's
# head start, 16 posting register scan, 8 chunked tree scan; FHIP_COLUMN_WALK=0: the leaf kernel by layers; FHIP_NO_ZREP=3: every slab of a frame without z).
# usage: gpurun -- 'bash tools/sweep_r05.sh'
O=gpurun_out/sweep_r05; mkdir -p $O
F='^RCCL\|^HIP ver\|^ROCm\|^Hostname\|^Librccl\|amdgpu.ids'
timeout 300 python -m pytest tests/test_prune2.py -m gpu -x -q 2>&1 | grep -v "$F" | tail -3
run() {
  echo "== $*"
  env "$@" timeout 120 python tools/p2stats.py 1024 2>&1 | grep -v "$F" | head -3 | cut -c1-200
  env "$@" python bench.py --no-cpu --no-general --steps 200 2>/dev/null | python -c "
import json,sys
r=json.loads(sys.stdin.read()); print('  bench', r['ms_per_step'], r['ms_per_step_median'], 'lone', r['frame_latency_ms'], r['roofline_timed_path']['kernel'], r['roofline_timed_path']['avg_launch_ms'])"
}
run A=0
run FHIP_DEBUG_BITS=32
run FHIP_NO_ZREP=3
run FHIP_NO_ZREP=3 FHIP_DEBUG_BITS=32
ROOT32_QUICK=1 python tools/root32.py 2>/dev/null | grep "lanes 0 no_inv 0 .* tiles     auto" > $O/root32_new.txt; cat $O/root32_new.txt
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/p_stats; timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_stats -o s -- python $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 2 --no-cpu --no-general > /dev/null 2>&1
cd $GRAFT_REPO_ROOT; find /tmp/p_stats -name "*kernel_stats.csv" -exec cp {} $O/kernel_stats_default.csv \;
python - <<'PY'
import csv
rows = list(csv.DictReader(open("gpurun_out/sweep_r05/kernel_stats_default.csv")))
fr = max(1, [int(r["Calls"]) for r in rows if r["Name"].startswith("k_finish3d")][0])
for r in rows[:18]:
    print(f"{r['Name'][:44]:44s} calls/frame {int(r['Calls'])/fr:5.2f} avg {float(r['AverageNs'])/1e3:7.1f} us")
PY
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_render_random.py tests/test_multi_gpu.py -m gpu -x -q -n 3 -k "render3d or random or octant or block or shard" 2>&1 | grep -v "$F" | tail -3
