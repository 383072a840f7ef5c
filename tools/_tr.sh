R=$PWD
timeout 1200 python -m pytest tests -m gpu -q 2>&1 | tail -5
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/tr; FHIP_NO_PIPELINE=1 rocprofv3 --kernel-trace --output-format csv -d /tmp/tr -o t -- python $R/bench.py --steps 2 --warmup 1 --no-cpu > /dev/null 2>&1
python $R/tools/trace_summary.py /tmp/tr 130 | grep "fh_columns" | tail -3
cd $R
for i in 1 2; do timeout 100 python bench.py --steps 40 --warmup 3 --no-cpu 2>/dev/null | python -c "
import json,sys,os;d=json.loads(sys.stdin.read());print(d['ms_per_step'], d['kernel_ms_per_frame'])"; done
