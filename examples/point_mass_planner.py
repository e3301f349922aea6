"""The planner of the reference's example/point_mass_planner.py (lines 8-66) written against optas_amd: the same builder calls; the
plot / animation part is out of scope.  Same plant as point_mass_mpc.py, but one plan over T = 45 knots from rest to rest around a
fixed obstacle, with the goal only on the last knot."""
import numpy as np

import optas_amd
from optas_amd.builder import OptimizationBuilder
from optas_amd.expr import sumsqr
from optas_amd.solver import HIPSolver


class Planner:
    def __init__(self, solver_options=None, build_only=False):
        dt, T = 0.1, 45
        obs, obs_rad, pm_radius = [0.0, 0.0], 0.2, 0.1
        point_mass = optas_amd.TaskModel("point_mass", 2, time_derivs=[0, 1], dlim={0: [-1.5, 1.5], 1: [-1, 1]})
        name = point_mass.get_name()
        builder = OptimizationBuilder(T, tasks=point_mass, derivs_align=True)
        init = builder.add_parameter("init", 2)
        goal = builder.add_parameter("goal", 2)
        builder.enforce_model_limits(name, time_deriv=0)
        builder.enforce_model_limits(name, time_deriv=1)
        builder.integrate_model_states(name, time_deriv=1, dt=dt)
        builder.fix_configuration(name, config=init)
        builder.fix_configuration(name, time_deriv=1)
        builder.add_equality_constraint("final_velocity", builder.get_model_state(name, -1, time_deriv=1))
        X = builder.get_model_states(name)
        for i in range(T):
            builder.add_geq_inequality_constraint(f"obs_avoid_{i}", sumsqr(obs - X[:, i]), (obs_rad + pm_radius) ** 2)
        builder.add_cost_term("final_state", sumsqr(goal - X[:, -1]))
        dX = builder.get_model_states(name, time_deriv=1)
        builder.add_cost_term("minimize_velocity", (0.01 / float(T)) * sumsqr(dX))
        builder.add_cost_term("minimize_acceleration", (0.005 / float(T)) * sumsqr((dX[:, 1:] - dX[:, :-1]) / dt))
        self.optimization = builder.build()
        self.T, self.dt, self.name, self.duration = T, dt, name, float(T - 1) * dt
        self.solver = None if build_only else HIPSolver(self.optimization).setup("hip_sqp", solver_options)

    def plan(self, init, goal, seed=None):
        """The script calls plan([-1, -1], [1, 1]): start, obstacle and goal on one line.  That instance is mirror-symmetric and the
        symmetric stationary point (stopping in front of the obstacle) is where an exactly symmetric Newton iteration stays; any seed
        with a lateral component decides the side, `seed` = (2, T) velocity guess."""
        self.solver.reset_parameters({"init": init, "goal": goal})
        if seed is not None:
            self.solver.reset_initial_seed({f"{self.name}/dy/x": seed})
        solution = self.solver.solve()
        return (self.solver.interpolate(solution[f"{self.name}/y"], self.duration), self.solver.interpolate(solution[f"{self.name}/dy"], self.duration),
                solution)


def main():
    planner = Planner()
    seed = np.zeros((2, planner.T))
    seed[0, 1:-1] = 0.05  # pass the obstacle on the +x side
    plan_y, plan_dy, sol = planner.plan([-1.0, -1.0], [1.0, 1.0], seed)
    print("did_solve", planner.solver.did_solve(), "iterations", planner.solver.number_of_iterations(), "f", planner.solver.stats()["f"][0])
    print("y(T) =", plan_y(planner.duration), "dy(T) =", plan_dy(planner.duration))
    return 0


if __name__ == "__main__":
    raise SystemExit(main())
