import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'tests'))
import numpy as np
from pympc_amd import fixtures
from test_gpu_parity import _stacked_batch
kws = [fixtures.random_lti(300 + i) for i in range(2)]
K = _stacked_batch(kws); K.setup()
print('setup ok', flush=True)
tr = K.run(int(sys.argv[1]) if len(sys.argv) > 1 else 2)
print(tr['u'][:, 0], tr['status'], tr['iter'])
