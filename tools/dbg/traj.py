import sys, os, json
sys.path.insert(0, os.getcwd())
import numpy as np, torch, tempfile
from tests.test_calculators_gpu import make_calc, REF_TRAINING
f=np.load('tests/golden/features_164x54.npz'); X=np.ascontiguousarray(f['X']); names=[str(s) for s in f['names']]
z=np.load('tools/dbg/f64traj.npz')
from deep_cartograph_amd import hip
for mode in ('split','native'):
    hip.set_gemm_mode(mode)
    tr=json.loads(json.dumps(REF_TRAINING)); tr['general']['max_epochs']=21; tr['early_stopping']['patience']=50
    calc=make_calc('deep_tica', tempfile.mkdtemp(), training=tr)
    calc.set_training_matrix(X.copy(), names)
    assert calc.train()
    v=np.array(calc.metrics['valid_loss'])
    print(mode,'eng-f64',np.array2string(np.abs(v-z['valid64']),precision=2))
    print(mode,'f32-f64',np.array2string(np.abs(z['valid32']-z['valid64']),precision=2))
    for i,(w,b) in enumerate(calc.cv['linears']):
        print(mode,'layer',i,'|W-W64|',np.abs(w-z[f'w{2*i}']).max(),'|b-b64|',np.abs(b-z[f'w{2*i+1}']).max())
