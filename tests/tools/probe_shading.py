import sys; sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tests')
import numpy as np, ctypes as C
import bevy_hikari_amd as hk
from bevy_hikari_amd import _ffi as F
from oracle_lib import oracle_api
rng = np.random.default_rng(0)
eng = hk.Engine(device=0)
def orc(op, x, y, n):
    out = np.empty(n, np.float32); fp = lambda a: a.ctypes.data_as(C.POINTER(F.f32))
    oracle_api().call("debug_math", None, op, fp(x), fp(y), fp(out), n); return out
def gpu(op, x, y, n):
    out = np.empty(n, np.float32); fp = lambda a: a.ctypes.data_as(C.POINTER(F.f32))
    eng.api.call("debug_math", eng.ctx, op, fp(x), fp(y), fp(out), n); return out
for op in (11, 12, 13):
    x = np.concatenate([np.array([-0.0, 0.0, -1e-30, 1e-30, np.nan, -np.inf, np.inf, 1.0, -1.0, 2.0, -2.0], np.float32), rng.normal(0, 1, 1000).astype(np.float32)])
    y = np.concatenate([np.array([1.0, 1.0, 1, 1, 1, 1, 1, 1, 1, 1, -0.0], np.float32), rng.normal(0, 1, 1000).astype(np.float32)])
    a, b = gpu(op, x, y, x.size), orc(op, x, y, x.size)
    bad = (a.view(np.uint32) != b.view(np.uint32)) & ~(np.isnan(a) & np.isnan(b))
    print("op", op, "mismatches", bad.sum(), [(x[i], y[i], a[i], b[i]) for i in np.nonzero(bad)[0][:8]])
n = 200000
x = rng.normal(0, 1, (n, 16)).astype(np.float32)
x[:, 9:12] = rng.uniform(0, 1, (n, 3)); x[:, 12:15] = rng.uniform(0, 300, (n, 3)); x[:, 15] = rng.choice([0.0, 1.0, 2.0], n)
# make N.L and N.V mostly positive
x[:, 6:9] = x[:, 3:6] + 0.8 * x[:, 6:9]; x[:, 0:3] = x[:, 3:6] + 0.8 * x[:, 0:3]
x = np.ascontiguousarray(x)
y = rng.choice([1.0, 0.5, 0.089, 0.3], n).astype(np.float32)
for op in (16, 17, 18, 19):
    a, b = gpu(op, x, y, n), orc(op, x, y, n)
    bad = (a.view(np.uint32) != b.view(np.uint32)) & ~(np.isnan(a) & np.isnan(b))
    print("op", op, "mismatches", bad.sum(), "of", n, [(a[i], b[i]) for i in np.nonzero(bad)[0][:5]])
    if bad.any():
        i = np.nonzero(bad)[0][0]; print("   inputs", x[i], y[i])
