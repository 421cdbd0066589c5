python -m pytest tests -m gpu -q -x 2>&1 | tail -3 > gpurun_out/k1_tests.log
for rep in 1 2; do
python scripts/lat_ab.py --lib build_abl/k1.so --batches 256,512,1024,2048 --backends bcr8 2>&1 | grep batch | sed 's/^/k1    /'
python scripts/lat_ab.py --batches 256,512,1024,2048 --backends bcr8 2>&1 | grep batch | sed 's/^/queue /'
done > gpurun_out/k1_ab.log
cat gpurun_out/k1_tests.log; cut -c1-150 gpurun_out/k1_ab.log
python scripts/diag_makespan.py 1024 20
