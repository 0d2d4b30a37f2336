#!/usr/bin/env python3
"""Known answers for the GA operators (monsoon_ga_offspring / monsoon_ga_select), by numpy and the host GA mirror themselves:

    np.random.seed(s); population of mu individuals; Population.generate_offspring()   (evo/population.py:75-89 as mirrored in
    monsoon_amd/population.py, whose seeded individuals are pinned to the reference's by tests/golden/population_seed42.npz)

-> tests/golden/ga_kat.npz: the stream state before and after, parents (w, sigma), the lambda children, plus a raw-stream case:
20 000 legacy_gauss values with the stream positions they leave behind (the accept / reject pattern of the polar method).
numpy's legacy stream is frozen (NEP 19); exp / log are the host's (glibc, numpy's SIMD exp): consumers compare values to a few ulp
and positions exactly."""
import os
import sys

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO)
from monsoon_amd.config import EvolutionaryConfig  # noqa: E402
from monsoon_amd.population import Population  # noqa: E402

out = {}
for tag, (mu, lam, seed) in {"a": (16, 1024, 42), "b": (128, 128, 7)}.items():
    cfg = EvolutionaryConfig(mu=mu, lambda_=lam, seed=seed)
    pop = Population(cfg)
    pop.initialize_population(10)
    np.random.normal(0, 1)           # leave a cached second normal behind: has_gauss = 1 going in
    st0 = np.random.get_state()
    kids = pop.generate_offspring()
    st1 = np.random.get_state()
    out[f"{tag}_cfg"] = np.array([mu, lam, 10], dtype=np.int64)
    out[f"{tag}_params"] = np.array([cfg.tau, cfg.tau_prime, cfg.min_sigma])
    out[f"{tag}_pw"] = np.stack([p.weights for p in pop.individuals])
    out[f"{tag}_ps"] = np.stack([p.sigmas for p in pop.individuals])
    out[f"{tag}_kw"] = np.stack([k.weights for k in kids])
    out[f"{tag}_ks"] = np.stack([k.sigmas for k in kids])
    for name, st in (("st0", st0), ("st1", st1)):
        out[f"{tag}_{name}_key"] = st[1].astype(np.uint32)
        out[f"{tag}_{name}_pos"] = np.array([st[2], st[3]], dtype=np.int64)
        out[f"{tag}_{name}_gauss"] = np.array([st[4]])
rs = np.random.RandomState(2026)
g = np.zeros(20000)
pos = np.zeros(20000, dtype=np.int64)
blocks = 0
last = rs.get_state()[2]
for i in range(20000):
    g[i] = rs.normal(0, 1)
    p = rs.get_state()[2]
    if p < last:
        blocks += 1
    last = p
    pos[i] = blocks * 624 + p
out["gauss_seed"] = np.array([2026], dtype=np.int64)
out["gauss_values"] = g
out["gauss_stream_pos"] = pos   # absolute outputs consumed after each value (624 = the initial position of a seeded stream)
path = os.path.join(REPO, "tests", "golden", "ga_kat.npz")
np.savez_compressed(path, **out)
print(path, {k: v.shape for k, v in out.items() if k.endswith("kw") or k.startswith("gauss")})
