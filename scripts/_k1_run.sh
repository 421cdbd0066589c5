cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
rm -rf /tmp/ks; ( cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ks -o s -- python $GRAFT_REPO_ROOT/scripts/setup_time.py 2>&1 | grep "setup " )
python - <<'PY'
import csv, glob
for f in glob.glob('/tmp/ks/**/*kernel_stats.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if 'setup' in r['Name'] or 'pack' in r['Name']:
            print('   ', r['Name'][:60], 'calls', r['Calls'], 'avg us %.1f' % (float(r['AverageNs']) / 1e3), 'min us %.1f' % (float(r['MinNs']) / 1e3), 'max us %.1f' % (float(r['MaxNs']) / 1e3))
PY
timeout 1200 python -m pytest tests -m gpu -q -x 2>&1 | tail -3
python bench.py --steps 20 --warmup 5 2>/dev/null | tail -n 1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['cold'])"
