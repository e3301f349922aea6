"""The reference's example/point_mass_mpc.py Controller (lines 88-175) written against optas_amd: the same
builder calls in the same order; CasADiSolver(...).setup("ipopt") becomes HIPSolver(...).setup("hip_sqp")."""
import numpy as np

import optas_amd
from optas_amd.builder import OptimizationBuilder
from optas_amd.expr import sumsqr
from optas_amd.solver import HIPSolver


class Controller:
    def __init__(self, solver_options=None, build_only=False):
        dt = 0.05  # time step
        obs_rad = 0.2  # obstacle radii
        pm_radius = 0.1  # point mass radii
        pm_dim = 2
        dlim = {0: [-1.5, 1.5], 1: [-1, 1]}  # pos/vel limits
        point_mass = optas_amd.TaskModel("point_mass", pm_dim, time_derivs=[0, 1], dlim=dlim)
        pm_name = point_mass.get_name()
        T = 20
        builder = OptimizationBuilder(T, tasks=point_mass, derivs_align=True)
        curr = builder.add_parameter("curr", 2)
        dcurr = builder.add_parameter("dcurr", 2)
        goal = builder.add_parameter("goal", 2, T)
        obs = builder.add_parameter("obs", 2, T)
        builder.enforce_model_limits(pm_name, time_deriv=0)
        builder.enforce_model_limits(pm_name, time_deriv=1)
        builder.integrate_model_states(pm_name, time_deriv=1, dt=dt)
        builder.fix_configuration(pm_name, config=curr)
        builder.fix_configuration(pm_name, config=dcurr, time_deriv=1)
        X = builder.get_model_states(pm_name)
        safe_dist_sq = (obs_rad + pm_radius) ** 2
        for i in range(T):
            dist_sq = sumsqr(obs[:, i] - X[:, i])
            builder.add_geq_inequality_constraint(f"obs_avoid_{i}", dist_sq, safe_dist_sq)
        builder.add_cost_term("optimal_path", sumsqr(goal - X))
        dX = builder.get_model_states(pm_name, time_deriv=1)
        w = 0.0025 / float(T)
        ddX = (dX[:, 1:] - dX[:, :-1]) / dt
        builder.add_cost_term("minimize_acceleration", w * sumsqr(ddX))
        self.optimization = builder.build()
        self.T, self.dt, self.pm_name = T, dt, pm_name
        self.duration = float(T - 1) * dt
        self.solution = None
        if not build_only:
            self.solver = HIPSolver(self.optimization).setup("hip_sqp", solver_options)

    def next_state(self, curr, dcurr, goal, obs):
        if self.solution is not None:
            self.solver.reset_initial_seed(self.solution)  # warm start (point_mass_mpc.py:157-158)
        params = {"curr": curr, "dcurr": dcurr, "goal": goal, "obs": obs}
        self.solver.reset_parameters(params)
        self.solution = self.solver.solve()
        if not self.solver.did_solve():
            raise RuntimeError("solver failed")
        plan_y = self.solver.interpolate(self.solution[f"{self.pm_name}/y"], self.duration)
        plan_dy = self.solver.interpolate(self.solution[f"{self.pm_name}/dy"], self.duration)
        return plan_y(2 * self.dt), plan_dy(2 * self.dt), plan_y, plan_dy


def obstacle_and_goal(t, curr, T=20, dt=0.05, ramp=0.032):
    obs = np.array([[0.15 * np.sin((t + dt * i) * np.pi - np.pi), 0.15 * np.cos((t + dt * i) * np.pi - np.pi) + 0.15] for i in range(T)]).T
    goal = np.array([[curr[0] + ramp * i, curr[1] + ramp * i] for i in range(T)]).T
    return obs, goal


def main():
    c = Controller()
    curr, dcurr = np.array([-0.45, -0.35]), np.array([0.6, 0.6])
    t = 2.0
    for tick in range(5):
        obs, goal = obstacle_and_goal(t, curr)
        curr, dcurr, _, _ = c.next_state(curr, dcurr, goal, obs)
        print(f"tick {tick}: y={curr}, dy={dcurr}, f={c.solver.stats()['f'][0]:.6f}, iterations={c.solver.number_of_iterations()}")
        t += 2 * c.dt
    return 0


if __name__ == "__main__":
    raise SystemExit(main())
