"""Problem builders shared by the CPU (emulator) and GPU parity tests."""
import numpy as np

from pink_amd import synthetic
from pink_amd.batch import BarrierTerm, DenseTaskTerm, DiagonalTaskTerm, pack_terms


def golden_case(golden, name):
    """Terms of one fixture in tests/golden/pink_build_ik.npz -> (packed batch, P, q, G, h)."""
    g = golden
    nv, root, dt = int(g[f"{name}/nv"]), int(g[f"{name}/root"]), float(g[f"{name}/dt"])
    J, e, cost = g[f"{name}/J"], g[f"{name}/e"], g[f"{name}/cost"]
    tasks = [
        DenseTaskTerm(J=J[i][None], e=e[i][None], cost=cost[i], gain=float(g[f"{name}/gain"][i]),
                      lm_damping=float(g[f"{name}/lm"][i]))
        for i in range(J.shape[0])
    ]
    tasks.append(DiagonalTaskTerm(col0=root, e=g[f"{name}/e_posture"][None], cost=float(g[f"{name}/posture_cost"])))
    q, q_min, q_max, v_max = g[f"{name}/q"], g[f"{name}/q_min"], g[f"{name}/q_max"], g[f"{name}/v_max"]
    lb = np.full(nv, -np.inf)
    ub = np.full(nv, np.inf)
    ci = g[f"{name}/config_limit_indices"]
    vi = g[f"{name}/velocity_limit_indices"]
    lb[ci] = 0.5 * (q_min - q)[ci]
    ub[ci] = 0.5 * (q_max - q)[ci]
    lb[vi] = np.maximum(lb[vi], -dt * v_max[vi])
    ub[vi] = np.minimum(ub[vi], dt * v_max[vi])
    barriers = []
    if f"{name}/barrier_J" in g:
        for i in range(g[f"{name}/barrier_J"].shape[0]):
            sd = g[f"{name}/barrier_dq_safe"][i][None] if f"{name}/barrier_dq_safe" in g else None
            barriers.append(BarrierTerm(J_h=g[f"{name}/barrier_J"][i][None], h=g[f"{name}/barrier_h"][i][None],
                                        gain=float(g[f"{name}/barrier_gain"][i]),
                                        safe_displacement_gain=float(g[f"{name}/barrier_safe_gain"][i]),
                                        safe_displacement=sd))
    eq = []
    for i in range(int(g[f"{name}/n_constraints"]) if f"{name}/n_constraints" in g else 0):
        # what pink_amd.solve_ik._equalities hands over: A = J, b = -gain e per constraint task
        eq.append((g[f"{name}/constraint{i}_J"][None],
                   (-float(g[f"{name}/constraint{i}_gain"]) * g[f"{name}/constraint{i}_e"])[None]))
    batch = pack_terms(nv, tasks, dt, 1e-12, boxes=[(lb[None], ub[None])], barriers=barriers, batch_size=1,
                       equality_rows=eq)
    return batch, g[f"{name}/P"], g[f"{name}/qvec"], g[f"{name}/G"], g[f"{name}/h"]


GOLDEN_NAMES = ["ur5", "draco3", "barrier", "equality", "safe"]


def golden_equalities(golden, name):
    """(A, b) the reference's build_ik produced for the fixture, or (None, None)."""
    if f"{name}/A" in golden:
        return golden[f"{name}/A"], golden[f"{name}/b"]
    return None, None


def random_case(nv, B, seed, Kd_tasks=2, md=0, diag=True, tight=0.05, root=0, lm=0.0, rank_deficient=False):
    """Generic random batch: returns (packed batch, pink-form dict for the oracle)."""
    rng = np.random.default_rng(seed)
    tasks, Js, es, costs, gains, lms, rows = [], [], [], [], [], [], [0]
    for t in range(Kd_tasks):
        k = int(rng.integers(1, 7))
        J = rng.normal(0, 0.5, size=(B, k, nv))
        if rank_deficient:
            J[:, :, nv // 2:] = 0.0
        e = 0.1 * rng.normal(size=(B, k))
        cost = rng.uniform(0.2, 3.0, size=k)
        gain = float(rng.uniform(0.3, 1.0))
        tasks.append(DenseTaskTerm(J=J, e=e, cost=cost, gain=gain, lm_damping=lm))
        Js.append(J), es.append(e), costs.append(cost), gains.append(gain), lms.append(lm)
        rows.append(rows[-1] + k)
    if diag:
        k = nv - root
        e = rng.uniform(-0.5, 0.5, size=(B, k))
        tasks.append(DiagonalTaskTerm(col0=root, e=e, cost=0.1, gain=1.0))
        Js.append(np.broadcast_to(np.eye(nv)[root:], (B, k, nv))), es.append(e)
        costs.append(np.full(k, 0.1)), gains.append(1.0), lms.append(0.0)
        rows.append(rows[-1] + k)
    lb = np.full((B, nv), -np.inf)
    ub = np.full((B, nv), np.inf)
    lb[:, root:] = -rng.uniform(0.002, tight, size=(B, nv - root))
    ub[:, root:] = rng.uniform(0.002, tight, size=(B, nv - root))
    # a few coordinates without any bound, a few one-sided
    free = rng.random(size=(B, nv)) < 0.1
    lb[free] = -np.inf
    one = rng.random(size=(B, nv)) < 0.1
    ub[one] = np.inf
    dense_rows = []
    G_blocks, h_blocks = [], []
    if md:
        G = rng.normal(0, 1.0, size=(B, md, nv))
        h = rng.uniform(0.0, 0.05, size=(B, md))
        dense_rows.append((G, h))
        G_blocks.append(G), h_blocks.append(h)
    batch = pack_terms(nv, tasks, 0.005, 1e-12, boxes=[(lb, ub)], dense_rows=dense_rows, batch_size=B)
    # pink form: box rows as +-e_i rows (only finite ones), then dense rows
    eye = np.eye(nv)
    Gb = np.concatenate([np.broadcast_to(eye, (B, nv, nv)), np.broadcast_to(-eye, (B, nv, nv))], axis=1)
    hb = np.concatenate([ub, -lb], axis=1)
    hb = np.where(np.isfinite(hb), hb, 1e30)  # "no bound": a row that can never be active
    G_all = np.concatenate([Gb] + G_blocks, axis=1)
    h_all = np.concatenate([hb] + h_blocks, axis=1)
    pf = dict(J=np.ascontiguousarray(np.concatenate(Js, axis=1)), e=np.concatenate(es, axis=1),
              cost=np.concatenate(costs), gain=np.array(gains), lm=np.array(lms), rows=np.array(rows, np.int32),
              damping=1e-12, G=np.ascontiguousarray(G_all), h=np.ascontiguousarray(h_all))
    return batch, pf


CONFIG_CASES = [
    ("ur5", "tight", "dense"), ("ur5", "kinematic", "kinematic"),
    ("draco3", "tight", "dense"), ("draco3", "kinematic", "kinematic"),
    ("jvrc", "tight", "dense"), ("jvrc", "kinematic", "kinematic"),
]


def config_case(name, bounds, jac, B, seed=None):
    t = synthetic.make_terms(name, B, bounds=bounds, jacobians=jac, seed=seed)
    return synthetic.pack(t), synthetic.pink_form(t)
