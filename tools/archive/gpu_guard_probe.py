import numpy as np, time, sys
sys.path.insert(0, "/root/repo")
import optas_amd
from optas_amd import _lib
from optas_amd.backend import FigureEightBackend
from optas_amd.models import RobotModel
from oracle.robot import OracleRobot
from oracle.structured import FoldedChain
from oracle.guarded import Guards, solve_free_al
from oracle.problems import dual_arm_offsets
QC = np.deg2rad([0, -30, 0, 90, 0, 30, 0])
links=['end_effector_ball','lwr_arm_7_link','lwr_arm_5_link','lwr_arm_6_link']
obs = np.array([[0.55,0.0,z] for z in (0.1,0.2,0.3,0.4,0.5,0.6)])
T=int(sys.argv[1]) if len(sys.argv)>1 else 50
B=int(sys.argv[2]) if len(sys.argv)>2 else 8
off = dual_arm_offsets(T)["l"].T; dt = 10.0/(T-1)
rm = RobotModel.builtin("kuka_lwr", time_derivs=[0,1], name="kukal"); rm.add_base_frame("global_world", xyz=[0.0,-0.25,0.0])
g = _lib.oh_guards(); g.limits=1
lo, up = rm.lower_actuated_joint_limits, rm.upper_actuated_joint_limits
for j in range(7): g.q_lo[j], g.q_up[j] = lo[j], up[j]
att = rm.link_attachments("end_effector_ball", links)
g.n_links=len(links); g.n_obstacles=len(obs)
for l,(k,o) in enumerate(att):
    g.link_joint[l]=k
    for i in range(3): g.link_offset[l][i]=o[i]
be = FigureEightBackend(rm.kinematic_chain("end_effector_ball"), T, dt, off, w_path=1.0, w_vel=0.01, max_iter=400, lock_orientation=False, fix_dq0=False, path_in_frame=False, guards=g)
rng=np.random.default_rng(0)
qc = QC + np.concatenate([np.zeros((1,7)), rng.uniform(-0.05,0.05,(B-1,7))])
par = np.concatenate([qc, np.full((B,4),0.15), np.tile(np.concatenate([np.concatenate([o,[0.1]]) for o in obs]),(B,1))],1)
x0 = np.concatenate([np.tile(qc,(1,T)), np.zeros((B,7*(T-1)))],1)
be.set_profiling(True)
res = be.solve(x0, par)
t=time.time(); res = be.solve(x0, par); wall=time.time()-t
print('timing', be.timing())
print("gpu", res.status[:8], res.iters[:8], res.f[:4], res.kkt[:4], "ms", be.timing()["solve_ms"], wall)
if len(sys.argv)>3: sys.exit(0)
r = OracleRobot("/root/repo/optas_amd/robots/kuka_lwr.kin.json", name="kukal"); r.add_base_frame("global_world", xyz=[0.0,-0.25,0.0])
ch = FoldedChain(r, "end_effector_ball")
G = Guards(lo=r.lower_actuated_joint_limits, up=r.upper_actuated_joint_limits, links=links, link_radii=np.full(4,0.15), obs_pos=obs, obs_radii=np.full(6,0.1))
for b in range(min(B,3)):
    s = solve_free_al(ch, T, dt, off, qc[b], G, Q0=np.tile(qc[b],(T,1)), rho0=10.0, exact=False, max_iter=400)
    Qg = res.x[b,:7*T].reshape(T,7)
    print("cpu", s["status"], s["iters"], s["f"], s["stat"], s["meas"], "dQ", np.abs(Qg-s["Q"]).max(), "df", res.f[b]-s["f"])
lam = be.multipliers(B)
print("lam", lam.shape, (lam>0).sum(axis=(1,2))[:4], "cpu active", (s["lam"]>0).sum())
