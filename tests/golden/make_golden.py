#!/usr/bin/env python3
"""Generates tests/golden/*.npz: inputs + expected iterates for short runs of every algorithm on the hot path.

PROVENANCE: these vectors are produced by the repo's own CPU restatement (oracle/), NOT by the Julia reference (which
cannot run in this image).  They freeze the restatement -- itself pinned by the reference's known-answer tests
(tests/test_oracle_pins.py) -- so that later edits to the oracle or to the kernels are caught as drift.
The literal 8 x 5 logistic data, labels and x_star are the reference's own fixture (test/test_logistic_l1.jl:12-29).

Run from the repo root:  python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import problems as P  # noqa: E402
from oracle import oracle as O  # noqa: E402


def run_case(name, loss, A, b, lam_f, L, lam_g, x0, seed):
    N, d = A.shape
    rng = np.random.default_rng(seed)
    p = O.Problem(loss, A, b, lam_f)
    g = O.Prox("l1", lam=lam_g)
    out = dict(loss=loss, A=A, b=b, lam_f=lam_f, L=L, lam_g=lam_g, x0=x0)
    # SVRG: init + 3 epochs of m = N draws
    gamma = 1.0 / (7 * L.max())
    idx = rng.integers(0, N, size=(3, N))
    av, z, zf, w = O.svrg_init(p, x0)
    out["svrg_gamma"], out["svrg_idx"], out["svrg_av0"] = gamma, idx, av.copy()
    for e in range(3):
        O.svrg_iterate(p, g, gamma, idx[e], False, av, z, zf, w)
    out["svrg_zfull"], out["svrg_w"], out["svrg_av"] = zf.copy(), w.copy(), av.copy()
    # SAGA and SAG: init + 5N steps
    for sag in (False, True):
        gam = 1.0 / ((16 if sag else 3) * L.max())
        sidx = rng.integers(0, N, size=5 * N)
        table, av, z = O.saga_init(p, g, gam, x0)
        key = "sag" if sag else "saga"
        out[f"{key}_gamma"], out[f"{key}_idx"], out[f"{key}_z0"] = gam, sidx, z.copy()
        O.saga_steps(p, g, gam, sag, sidx, table, av, z)
        out[f"{key}_z"], out[f"{key}_av"], out[f"{key}_table"] = z.copy(), av.copy(), table.copy()
    # Finito (cyclic, batch 2: the first step uses batch #2, Finito_basic.jl:99) and LFinito (identity order)
    gam_i = (0.999 * N / L).astype(A.dtype)
    nb = -(-N // 2)
    static = [np.arange(2 * j, min(2 * j + 2, N), dtype=np.int64) for j in range(nb)]
    table, av, z, hg = O.finito_init(p, g, gam_i, x0)
    out["finito_gam"], out["finito_hat_gamma"], out["finito_z0"] = gam_i, hg, z.copy()
    batches = [static[(t + 1) % nb] for t in range(3 * nb)]
    O.finito_steps(p, g, gam_i, hg, batches, table, av, z)
    out["finito_z"], out["finito_av"], out["finito_table"] = z.copy(), av.copy(), table.copy()
    av, z, zf, hg = O.lfinito_init(p, gam_i, x0)
    for _ in range(3):
        O.lfinito_iterate(p, g, gam_i, hg, static, av, z, zf)
    out["lfinito_z"], out["lfinito_zfull"], out["lfinito_av"] = z.copy(), zf.copy(), av.copy()
    np.savez(os.path.join(HERE, f"{name}.npz"), **out)
    print("wrote", name, {k: np.shape(v) for k, v in out.items() if k.endswith("_z")})


def main():
    only = set(sys.argv[1:])   # optional: names of the cases to (re)generate; default all
    want = lambda name: not only or name in only  # noqa: E731
    if want("logistic_l1_reference_fixture_f64"):
        A, y, L, lam, x0, x_star = P.logistic_fixture(np.float64)
        run_case("logistic_l1_reference_fixture_f64", "logistic", A, y, 1.0, L, lam, x0, seed=1)
    if want("lasso_known_answer_f64"):
        A, b, L, lam, x0, x_star, f_star = P.lasso_known_answer(dtype=np.float64)
        run_case("lasso_known_answer_f64", "ls", A, b, float(A.shape[0]), L, lam, x0, seed=2)
    if want("lasso_known_answer_f32"):
        A, b, L, lam, x0, x_star, f_star = P.lasso_known_answer(dtype=np.float32)
        run_case("lasso_known_answer_f32", "ls", A, b, float(A.shape[0]), L, lam, x0, seed=2)
    if want("lasso_d1024_f64"):
        # a BASELINE-shaped (d = 1024) instance, small N: exercises the fast sweep and the LDS-DMA chains
        A, b, x = P.synthetic("ls", 16, 1024, np.float64, seed=3)
        L = 16.0 * np.sum(A * A, axis=1)
        run_case("lasso_d1024_f64", "ls", A, b, 16.0, L, 0.01, np.zeros(1024), seed=3)
    if want("lasso_d1000_f64"):
        # a row length outside the wave-per-row shapes: the masked workgroup-per-row kernel and the masked LDS-DMA chains
        A, b, x = P.synthetic("ls", 12, 1000, np.float64, seed=4)
        L = 12.0 * np.sum(A * A, axis=1)
        run_case("lasso_d1000_f64", "ls", A, b, 12.0, L, 0.01, np.zeros(1000), seed=4)
    if want("logistic_d50_f32"):
        # rows of 200 bytes: several rows per wave in the sweeps, tiny-row chains
        A, y, x = P.synthetic("logistic", 40, 50, np.float32, seed=5)
        L = (0.25 * np.sum(A.astype(np.float64) ** 2, axis=1)).astype(np.float32)
        run_case("logistic_d50_f32", "logistic", A, y, 1.0, L, 0.01, np.zeros(50, np.float32), seed=5)


if __name__ == "__main__":
    main()
