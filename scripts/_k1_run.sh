cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
for lib in ${LIBS:-pympc_amd/libmpcqp_hip.so}; do
echo "--- $lib"
rm -rf /tmp/ks; ( cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ks -o s -- python $GRAFT_REPO_ROOT/scripts/with_lib.py $GRAFT_REPO_ROOT/$lib $GRAFT_REPO_ROOT/scripts/setup_only.py 2>&1 | grep "setup " | tail -1 )
python - <<'PY'
import csv, glob
for f in glob.glob('/tmp/ks/**/*kernel_stats.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if 'setup' in r['Name']:
            print('   ', r['Name'][:60], 'calls', r['Calls'], 'avg us %.1f' % (float(r['AverageNs']) / 1e3), 'min us %.1f' % (float(r['MinNs']) / 1e3))
PY
done
