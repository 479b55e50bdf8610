"""The fixed member set behind the five-equation golden fixtures (shared by the two generator scripts and the tests):
16 members of the shard-computable Latin hypercube plus 8 corners of the perturbation box, for the CO2-only and the
multi-gas parameter sets, on the frozen 750-step RCP-like emissions.  Data only — no model code here."""
import numpy as np

from fiveeqscm_amd import emissions, params

N_LHS = 16
# stored steps: the first years, every tenth step, the emission peak / sign change, the last step
STEPS = sorted(set([0, 1, 2] + list(range(24, 750, 25)) + [282, 400, 749]))
# corners of the box (r0, rC, rT scale factors; TCR; ECS): all-low, all-high, weakest and strongest carbon-cycle
# feedback at high / low climate sensitivity, and mixed ones
CORNERS = [
    (0.8, 0.5, 0.5, 1.0, 1.5), (1.2, 1.5, 1.5, 2.5, 4.5), (1.2, 1.5, 1.5, 1.0, 1.5), (0.8, 0.5, 0.5, 2.5, 4.5),
    (1.2, 0.5, 1.5, 2.5, 4.5), (0.8, 1.5, 0.5, 1.0, 4.5), (1.2, 1.5, 0.5, 2.272, 2.5), (0.8, 0.5, 1.5, 1.0, 1.1),
]


def members(kind):
    """Parameter dict for the 24 golden members of parameter set `kind` ('co2' | 'multigas')."""
    base = params.default_params(kind)
    G = params.n_gas_of(base)
    p = params.sample_ensemble_shard(base, N_LHS)
    out = dict(base)
    F2x = params.forcing_2x(base)
    cols = {k: [p[k][:, i] for i in range(N_LHS)] for k in ("r0", "rC", "rT", "q")}
    for s0, sC, sT, tcr, ecs in CORNERS:
        cols["r0"].append(np.asarray(base["r0"], dtype=np.float64).reshape(G) * s0)
        cols["rC"].append(np.asarray(base["rC"], dtype=np.float64).reshape(G) * sC)
        cols["rT"].append(np.asarray(base["rT"], dtype=np.float64).reshape(G) * sT)
        cols["q"].append(params.k_q(tcr, ecs, base["d"], F2x))
    for k in cols:
        out[k] = np.stack(cols[k], axis=1)
    return out, N_LHS + len(CORNERS)


def scenario(kind):
    return emissions.rcp_like_emissions(750, 1 if kind == "co2" else 3)
