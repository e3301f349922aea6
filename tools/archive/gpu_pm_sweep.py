"""Development probe: point-mass tick at several batch sizes, thread kernel against the wavefront-per-plant kernel (OH_PM_WAVE_MAX)."""
import os, sys
import numpy as np
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
from optas_amd import _lib
from optas_amd.backend import PointMassBackend
T = 20
rng = np.random.default_rng(20260927)
obs = np.array([[0.15 * np.sin(np.pi * (0.05 * t) - np.pi), 0.15 * np.cos(np.pi * (0.05 * t) - np.pi) + 0.15] for t in range(T)])
for B in (1024, 4096, 8192, 16384, 32768, 65536):
    P = np.zeros((B, 4 + 4 * T))
    c = rng.uniform(-1.2, 1.2, (B, 2))
    far = np.linalg.norm(c - obs[0], axis=1) > 0.35
    c[~far] += 0.8
    P[:, :2] = c
    goal = np.clip(c[:, None, :] + (1 - c[:, None, :]) * (np.arange(T) / (T - 1.0))[None, :, None], -1.5, 1.5)
    P[:, 4 : 4 + 2 * T] = goal.reshape(B, -1)
    P[:, 4 + 2 * T :] = np.tile(obs.reshape(-1), (B, 1))
    x0 = np.zeros((B, 4 * T))
    out = []
    for mode in ("0", "1000000"):
        os.environ["OH_PM_WAVE_MAX"] = mode
        be = PointMassBackend(tol=1e-8)
        bufs = [_lib.DeviceBuffer(a.nbytes) for a in (x0, P)]
        bufs[0].upload(x0); bufs[1].upload(P)
        d = [_lib.DeviceBuffer(x0.nbytes), _lib.DeviceBuffer(8 * B), _lib.DeviceBuffer(24 * B), _lib.DeviceBuffer(4 * B), _lib.DeviceBuffer(4 * B)]
        ms = []
        for _ in range(4):
            be.solve_device(B, bufs[0], bufs[1], *d)
            ms.append(be.solve_ms())
        st = d[4].download(np.int32, (B,))
        out.append((float(np.median(ms[1:])), float((st == 0).mean())))
        for b in bufs + d:
            b.free()
        be.close()
    print(f"B={B}: thread kernel {out[0][0]:.2f} ms, wave kernel {out[1][0]:.2f} ms (converged {out[0][1]:.4f} / {out[1][1]:.4f})")
