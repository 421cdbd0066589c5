import os, sys, warnings
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np
from test_gpu_group import _ctrl, SHAPES
shape = SHAPES[0]
for grp in (sys.argv[1],):
    os.environ['MPCQP_GROUP'] = grp
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        K, Ko = _ctrl(shape), _ctrl(shape, oracle=True)
        print('setup gpu', flush=True); K.setup(); print('done', flush=True); Ko.setup()
    print('GROUP', grp, 'gpu', K.res.info.status, K.res.info.iter, K.res.info.rho_updates, 'rho', K.prob.batch_problem.infos()[0].rho, '| oracle', Ko.res.info.iter, Ko.res.info.rho_updates, 'rho_estimate', Ko.res.info.rho_estimate, 'rho now', Ko.prob.iterate_state()[3])
    print('   u gpu', K.output(), 'oracle', Ko.output())
