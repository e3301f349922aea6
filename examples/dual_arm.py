"""The reference's example/dual_arm.py DualKukaPlanner.setup_solver (lines 17-129) written against optas_amd: the same
builder calls in the same order.  The CasADi loop that assembles the piecewise-linear path (:96-113) is evaluated on
numpy offsets and added to the symbolic start position."""
import numpy as np

import optas_amd
from optas_amd.builder import OptimizationBuilder
from optas_amd.expr import sumsqr
from optas_amd.solver import HIPSolver

kukal_base_position = [0.0, -0.25, 0.0]
kukar_base_position = [0.0, 0.25, 0.0]


def _setup_kuka_model(name, base_position):
    model = optas_amd.RobotModel.builtin("kuka_lwr", time_derivs=[0, 1], name=name)
    model.add_base_frame("global_world", xyz=base_position)
    return model


def path_offsets(T, d1, d2):
    off = np.zeros((3, T))
    for i in range(T):
        alpha_ = float(i) / float(T - 1)
        if alpha_ < 0.4:
            off[:, i] = (alpha_ / 0.4) * np.asarray(d1)
        elif 0.4 <= alpha_ < 0.5:
            off[:, i] = d1
        else:
            off[:, i] = np.asarray(d1) + ((alpha_ - 0.5) / 0.5) * np.asarray(d2)
    return off


SPHERE_LINKS = ["end_effector_ball", "lwr_arm_7_link", "lwr_arm_5_link", "lwr_arm_6_link"]  # sphere_collision_avoidance.py:95
N_OBSTACLES = 6  # sphere_collision_avoidance.py:46-51


def obstacle_parameters(link_radius=0.15, obstacle_radius=0.1):
    """Parameter values of the synthetic config 4: a column of six spheres at x = 0.55, y = 0, z = 0.1 ... 0.6
    between the arms (sphere_collision_avoidance.py:46-53)."""
    p = {}
    for arm in ("kukal", "kukar"):
        for ln in SPHERE_LINKS:
            p[f"{arm}_{ln}_radii"] = link_radius
        for i in range(N_OBSTACLES):
            p[f"{arm}_obs{i}_position"] = np.array([0.55, 0.0, 0.1 * (i + 1)])
            p[f"{arm}_obs{i}_radii"] = obstacle_radius
    return p


NOMINAL_QC = np.deg2rad([0, -30, 0, 90, 0, 30, 0])  # dual_arm.py:185


def initial_clearance(arm, qc, link_radius=0.15, obstacle_radius=0.1):
    """Smallest sphere-clearance row of knot 0, ||c_l(qc) - o_j||^2 - (r_l + r_j)^2 over the sphere links and obstacles of the synthetic
    config 4, for each row of qc (B, ndof).  q_0 = qc is pinned by fix_configuration (builder.py:525-539), so these rows are constants of the
    instance: a negative one means the NLP as posed has no feasible point."""
    qc = np.atleast_2d(np.asarray(qc, dtype=np.float64))
    obs = np.array([[0.55, 0.0, 0.1 * (i + 1)] for i in range(N_OBSTACLES)])
    worst = np.full(len(qc), np.inf)
    for ln in SPHERE_LINKS:
        c = np.asarray(arm.get_global_link_position(ln, qc.T)).T  # (B, 3)
        d2 = ((c[:, None, :] - obs[None]) ** 2).sum(-1)
        worst = np.minimum(worst, (d2 - (link_radius + obstacle_radius) ** 2).min(1))
    return worst


def draw_feasible_configurations(rng, B, arm, link_radius=0.15, obstacle_radius=0.1, spread=0.1, margin=0.0):
    """SURVEY 8(d) C4's perturbed initial configurations qc = nominal + U(-spread, spread)^7, drawn by rejection so that every instance is feasible
    as posed (the way SURVEY C3 rejects point-mass starts inside the obstacle): at link radius 0.15 the nominal configuration has 1.6 mm of
    clearance and about four draws in five pin q_0 inside one."""
    out = np.empty((0, arm.ndof))
    while len(out) < B:
        cand = NOMINAL_QC + rng.uniform(-spread, spread, (max(64, 6 * (B - len(out))), arm.ndof))
        out = np.concatenate([out, cand[initial_clearance(arm, cand, link_radius, obstacle_radius) > margin]])
    return np.ascontiguousarray(out[:B])


def setup_solver(T=50, Tmax=10.0, solver_options=None, build_only=False, limits=False, collision=False, velocity_limits=None):
    link_ee = "end_effector_ball"
    t = np.linspace(0, Tmax, T)
    dt = float(t[1] - t[0])
    kukal = _setup_kuka_model("kukal", kukal_base_position)
    kukar = _setup_kuka_model("kukar", kukar_base_position)
    kukal_name, kukar_name = kukal.get_name(), kukar.get_name()
    builder = OptimizationBuilder(T=T, robots=[kukal, kukar])
    qcl = builder.add_parameter("qcl", kukal.ndof)
    qcr = builder.add_parameter("qcr", kukar.ndof)
    builder.fix_configuration(kukal_name, qcl)
    builder.fix_configuration(kukar_name, qcr)
    builder.integrate_model_states(kukal_name, time_deriv=1, dt=dt)
    builder.integrate_model_states(kukar_name, time_deriv=1, dt=dt)
    Ql, Qr = builder.get_model_states(kukal_name), builder.get_model_states(kukar_name)
    ee_pos_pathl = kukal.get_global_link_position(link_ee, Ql)
    ee_pos_pathr = kukar.get_global_link_position(link_ee, Qr)
    dQl = builder.get_model_states(kukal_name, time_deriv=1)
    dQr = builder.get_model_states(kukar_name, time_deriv=1)
    w_dq = 0.01
    builder.add_cost_term("kukal_min_join_vel", w_dq * sumsqr(dQl))
    builder.add_cost_term("kukar_min_join_vel", w_dq * sumsqr(dQr))
    pos0l = kukal.get_global_link_position(link_ee, qcl)
    pos0r = kukar.get_global_link_position(link_ee, qcr)
    path_eel = pos0l + path_offsets(T, [-0.1, 0.1, -0.2], [0.0, 0.0, 0.3])
    path_eer = pos0r + path_offsets(T, [-0.1, -0.1, -0.2], [0.0, 0.0, 0.3])
    builder.add_cost_term("ee_pos_pathl", sumsqr(ee_pos_pathl - path_eel))
    builder.add_cost_term("ee_pos_pathr", sumsqr(ee_pos_pathr - path_eer))
    # synthetic extensions of BASELINE config 4 (not in the shipped script): joint limits and sphere clearances per arm.
    # The reference names link-radius and obstacle parameters without the robot name, so the second arm would collide
    # on them (KeyError); hence per-arm obstacle names and the link_radii_prefix extension.
    if limits:
        builder.enforce_model_limits(kukal_name)
        builder.enforce_model_limits(kukar_name)
    if velocity_limits is not None:  # (lo, up) per joint, or True: the models' own (enforce_model_limits(name, time_deriv=1), builder.py:471-509)
        for arm in (kukal_name, kukar_name):
            if velocity_limits is True:
                builder.enforce_model_limits(arm, time_deriv=1)
            else:
                builder.enforce_model_limits(arm, time_deriv=1, lo=velocity_limits[0], up=velocity_limits[1])
    if collision:
        for arm in (kukal_name, kukar_name):
            builder.sphere_collision_avoidance_constraints(arm, [f"{arm}_obs{i}" for i in range(N_OBSTACLES)], link_names=SPHERE_LINKS,
                                                           link_radii_prefix=arm + "_")
    optimization = builder.build()
    if build_only:
        return (kukal, kukar), optimization
    return (kukal, kukar), HIPSolver(optimization).setup("hip_sqp", solver_options)


def main():
    (kukal, kukar), solver = setup_solver()
    qc = optas_amd.deg2rad([0, -30, 0, 90, 0, 30, 0])  # dual_arm.py:185
    solver.reset_parameters({"qcl": qc, "qcr": qc})  # the reference never sets a seed: zeros (solver.py:76)
    sol = solver.solve()
    print("did_solve", solver.did_solve(), "iterations", solver.number_of_iterations(), "f", solver.stats()["f"][0])
    print("kukal q(T-1) =", sol["kukal/q"][:, -1])
    return 0


if __name__ == "__main__":
    raise SystemExit(main())
