import sys, numpy as np
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tests')
from conftest import SEED
from optas_amd.backend import TorqueBackend
from optas_amd.models import RobotModel
from examples.torque_mpc import figure_eight_goal
LINK="lbr_link_ee"; W=dict(w_path=1000.0,w_vel=0.1,w_tau=1e-4); QC=np.deg2rad([0,30,0,-90,0,-30,0])
T,B=30,512
robot=RobotModel.builtin("med7")
rng=np.random.default_rng(SEED+3)
qc=QC+rng.uniform(-0.1,0.1,(B,7))
goal=np.stack([figure_eight_goal(robot,LINK,q,T,0.1).T for q in qc])
p=np.concatenate([qc,np.zeros((B,7)),goal.reshape(B,-1)],1)
x0=np.zeros((B,4*7*T)); x0[:,:7*T]=np.tile(qc,(1,T))
for vm in (0.4,0.5):
    be=TorqueBackend(robot.kinematic_chain(LINK),robot.dynamics_tables(),T=T,dt=0.1,tau_lo=-58.0,tau_up=58.0,dq_lo=-vm,dq_up=vm,max_iter=1000,**W)
    r=be.solve(x0,p); ok=r.status==0
    print(vm,"conv",ok.mean(),"iters p50/p90/max",np.median(r.iters),np.percentile(r.iters,90),r.iters.max(),"bad kkt",r.kkt[~ok][:6], "ms", be.timing()["solve_ms"])
    be.close()
