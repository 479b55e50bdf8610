import sys, ctypes, numpy as np, torch
sys.path.insert(0, '.')
from fiveeqscm_amd import _capi
rng = np.random.default_rng(7)
x = -np.concatenate([10.0 ** rng.uniform(-300, 2.9, 2_000_000), rng.uniform(0, 2, 2_000_000), rng.uniform(0, 40, 2_000_000)])
y = np.concatenate([rng.uniform(-700, 700, 2_000_000), rng.uniform(-12, 12, 2_000_000)])
for path in sys.argv[1:]:
    lib = _capi.load(path)
    for op, arr, ref in ((0, x, np.expm1), (1, y, np.exp)):
        xd = torch.from_numpy(arr).cuda(); yd = torch.empty_like(xd)
        lib.fiveeq_math_probe_f64(op, xd.numel(), ctypes.c_void_p(xd.data_ptr()), ctypes.c_void_p(yd.data_ptr()), None)
        torch.cuda.synchronize()
        got = yd.cpu().numpy(); want = ref(arr)
        u = np.abs(got - want) / np.spacing(np.abs(want))
        u = u[np.isfinite(u)]
        print(path.split('/')[-1], 'op', op, 'max ulp', u.max(), 'mean', u.mean(), '>1ulp frac', (u > 1).mean())
