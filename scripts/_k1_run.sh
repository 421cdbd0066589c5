cd $GRAFT_REPO_ROOT
for rep in 1 2; do
python scripts/lat_ab.py --lib build_abl/prev.so --batches 128,1024 --backends bcr8 2>&1 | grep batch | sed 's/^/prev /' | cut -c1-140
python scripts/lat_ab.py --batches 128,1024 --backends bcr8 2>&1 | grep batch | sed 's/^/new  /' | cut -c1-140
done
python scripts/with_lib.py build_abl/prev.so scripts/latency_single.py 2>&1 | grep "drop-in"
python scripts/latency_single.py 2>&1 | grep "drop-in"
python scripts/with_lib.py build_abl/prev.so scripts/latency_single.py 2>&1 | grep "drop-in"
python scripts/latency_single.py 2>&1 | grep "drop-in"
