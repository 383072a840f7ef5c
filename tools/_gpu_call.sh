for K in 60 100 60 100; do
python bench.py --no-cpu --steps $K 2>/dev/null | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read()); g=d['general']
print('K', d['steps'], 'default mean', round(d['ms_per_step'],4), 'median', round(d['ms_per_step_median'],4), '| general mean', round(g['ms_per_step'],4), 'median', round(g['ms_per_step_median'],4))"
done
