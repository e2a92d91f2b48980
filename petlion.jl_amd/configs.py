"""Deterministic synthetic inputs of the BASELINE.json configurations, shared by bench.py, tools/ and the tests.

RNG = counter-based splitmix64(seed ^ cell * 0x9E3779B97F4A7C15 ^ k) -> u in [0, 1): a cell's parameters depend only on (seed, global cell index, k),
so every rank of a multi-GPU run can build its own shard without communication and any subset of an ensemble is reproducible on its own.
"""
from __future__ import annotations

import numpy as np

SWEEP_KEYS = ("D_sp", "D_sn", "D_p", "D_s", "D_n", "k_p", "k_n")      # config C4 / C5: seven log-uniform factors 2^(2u-1)


def splitmix_u01(seed, cells, k):
    """u[cell] in [0, 1) for an array of global cell indices"""
    x = (np.uint64(seed) ^ (np.asarray(cells, dtype=np.uint64) * np.uint64(0x9E3779B97F4A7C15)) ^ np.uint64(k))
    with np.errstate(over="ignore"):
        x = x + np.uint64(0x9E3779B97F4A7C15)
        x = (x ^ (x >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        x = (x ^ (x >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        x = x ^ (x >> np.uint64(31))
    return (x >> np.uint64(11)).astype(np.float64) / float(1 << 53)


def _matrix(p, cells, over):
    Th = np.tile(p.theta_vector(), (len(cells), 1))
    for key, v in over.items():
        Th[:, p.θ_keys.index(key)] = v
    return Th


def sweep_theta(p, cells, seed):
    """the seven-parameter jitter of C4 / C5: key j of SWEEP_KEYS is multiplied by 2^(2 u(seed, cell, j) - 1).  Keys the model does not have
    (the NMC system has no D_p, D_s, D_n: its D_eff(c_e, T) is a closure, reference src/params.jl:407) are skipped, the others keep their index j."""
    cells = np.asarray(cells)
    return _matrix(p, cells, {key: p.θ[key] * 2.0 ** (2 * splitmix_u01(seed, cells, j) - 1) for j, key in enumerate(SWEEP_KEYS) if key in p.θ_keys})


def c2(p, n):
    """C2: n identical LCO cells, 1C CC discharge from SOC 1 to the stop condition"""
    return dict(name="C2", theta=_matrix(p, np.arange(n), {}), protocol=[{"I": -1.0}], SOC=1.0, max_points=256)


def c3(p, n, first=0):
    """C3: LCO with temperature = true, CC-CT-CV fast charge of examples/fast_charging_CC-CT-CV.ipynb (cells 5-13), cells differ by
    T_amb = 298.15 + 5 (u0 - 0.5) K and h_cell = 2^(2 u1 - 1), seed 3"""
    cells = first + np.arange(n)
    Th = _matrix(p, cells, {"T_amb": 298.15 + 5 * (splitmix_u01(3, cells, 0) - 0.5), "h_cell": 2.0 ** (2 * splitmix_u01(3, cells, 1) - 1)})
    kw = dict(T_max=313.15, V_max=4.1, I_max=4.0, I_min=1 / 20)
    return dict(name="C3", theta=Th, protocol=[dict(I=4.0, **kw), dict(dT="hold", **kw), dict(V="hold", **kw)], SOC=0.0, max_points=512)


def c4(p, n, first=0):
    """C4: LCO isothermal 1C discharge, seven-parameter log-uniform jitter, seed 4 (65 536 cells in total, 8 192 per GPU on 8 GPUs)"""
    return dict(name="C4", theta=sweep_theta(p, first + np.arange(n), 4), protocol=[{"I": -1.0}], SOC=1.0, max_points=256)


def c5(p, n, first=0, pulses=20):
    """C5: NMC + SEI aging, GITT of examples/GITT.ipynb:64-73: SOC0 = 0, 20 x {1C for 180 s ; rest 7200 s}, seven-parameter jitter, seed 5"""
    proto = []
    for _ in range(pulses):
        proto += [{"I": 1.0, "tf": 180.0}, {"I": "rest", "tf": 7200.0}]
    return dict(name="C5", theta=sweep_theta(p, first + np.arange(n), 5), protocol=proto, SOC=0.0, max_points=4096)
