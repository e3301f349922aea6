import os, sys, numpy as np
sys.path.insert(0, os.getcwd())
from examples.figure_eight_plan import setup_solver
QC0 = np.deg2rad([0, 30, 0, -90, 0, -30, 0])
B = 640
rng = np.random.default_rng(20260927 + 43)
qcs = QC0[None] + rng.uniform(-0.08, 0.08, (B, 7))
seeds = np.stack([np.tile(q.reshape(-1, 1), (1, 50)) for q in qcs])
out = {}
for sc in ("0", "2048"):
  os.environ["OH_SPARSE_CHECK_BELOW"] = sc
  for mode in ("0", "1"):
    os.environ["OH_COMPACTION"] = mode
    kuka, solver = setup_solver(velocity_limits=True, solver_options={"max_iter": 600, "tol": 1e-7})
    solver.reset_parameters_batch({"qc": qcs}); solver.reset_initial_seed_batch({"kuka/q/x": seeds})
    solver.solve_batch(stacked=True)
    st = solver.stats(); be = solver.backend
    out[(sc,mode)] = (st["f"].copy(), st["iterations"].copy(), st["status"].copy(), be.timing()["compactions"], be.timing()["iterations_launched"])
    be.close()
for sc in ("0","2048"):
    f0,i0,s0,c0,l0 = out[(sc,"0")]; f1,i1,s1,c1,l1 = out[(sc,"1")]
    d=np.abs(i0.astype(int)-i1)
    print(sc, 'compactions',c0,c1,'launched',l0,l1,'status ok',(s0==0).all(),(s1==0).all(),'within tol',(d<=np.maximum(3,i0//4)).mean(),'max diff',d.max(),'f diff',np.abs(f0-f1).max())
f0,i0,*_=out[("0","0")]; f1,i1,*_=out[("2048","0")]
print('no-compaction runs, sparse vs dense check: identical iters', np.array_equal(i0,i1), 'f', np.abs(f0-f1).max())
