import ctypes, sys, os
import numpy as np, torch
sys.path.insert(0, os.getcwd())
from fiveeqscm_amd import emissions, params
from fiveeqscm_amd.engine import EnsembleEngine
N = int(sys.argv[1]); steps = 200
base = params.sample_ensemble(params.default_params("multigas"), 65536)
p = dict(base)
for k in ("r0","rC","rT","q"): p[k] = np.tile(base[k], (1, -(-N//65536)))[:, :N]
E = emissions.rcp_like_emissions(steps+20, 3)
eng = EnsembleEngine(p, N, E, device="cuda:0")
lib = eng.lib; w = 8
def run_chunked(t0, t1, chunk):
    for c0 in range(0, N, chunk):
        n = min(chunk, N - c0)
        off = lambda t: ctypes.c_void_p(t.data_ptr() + c0 * w)
        rc = lib.fiveeq_run_f64(ctypes.byref(eng.model), n, N, ctypes.c_void_p(eng.drive.data_ptr()), eng.n_steps, t0, t1,
                                off(eng.r), off(eng.q), off(eng.R), off(eng.S), off(eng.C), off(eng.T), eng.n_rows, None,
                                ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
        assert rc == 0
def timeit(fn):
    fn(0, 20); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); fn(20, 20+steps); e1.record(); e1.synchronize()
    return e0.elapsed_time(e1)*1e3/steps
ref = timeit(lambda a,b: eng.run(a,b)); Rref = eng.R.clone(); Tref = eng.T.clone()
print(f"N={N} whole: {ref:.1f} us/step  {N/ref/1e3:.2f} G/s")
for chunk in (1<<19, 1<<20, 1<<21, 3<<19):
    eng.reset_state()
    t = timeit(lambda a,b: run_chunked(a,b,chunk))
    ok = torch.equal(eng.R, Rref) and torch.equal(eng.T[:220], Tref[:220])
    print(f"  chunk {chunk}: {t:.1f} us/step  {N/t/1e3:.2f} G/s  identical={ok}")
