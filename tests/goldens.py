"""Helpers shared by the parity tests: load a golden fixture (tests/golden/*.npz, produced by the
reference through tests/golden/make_golden.py) and rebuild its inputs."""
import hashlib
import os

import numpy as np

from oracle import pyoracle as po

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")

SMALL = ["iso64x4", "iso64x4_alpha3_2", "iso100x7", "iso64x4_f32", "dense96x16", "dense80x5_f32", "rosen80x8", "rosen200x33",
         "skewed320x2"]
DIGEST = ["iso600x130", "c2_16384x32", "c3_4096x32"]
# next row f3: Mover::DifferentialEvolution
DE_SMALL = ["de_iso64x4", "de_iso100x7", "de_iso14x3", "de_rosen80x8", "de_dense96x16", "de_dense80x5_f32"]
DE_DIGEST = ["de_c2_16384x32"]


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


class Golden:
    def __init__(self, name):
        self.name = name
        z = np.load(os.path.join(GOLDEN_DIR, name + ".npz"))
        self.z = z
        self.W, self.D = int(z["W"]), int(z["D"])
        self.calc, self.dtype = int(z["calc"]), int(z["dtype"])
        self.seed, self.slicing, self.steps = int(z["seed"]), int(z["slicing"]), int(z["steps"])
        self.params = z["params"] if z["params"].size else None
        self.alpha = tuple(int(v) for v in z["alpha"]) if "alpha" in z.files else (2, 1)
        self.mover = int(z["mover"]) if "mover" in z.files else 0
        self.accepted_per_call = z["accepted_per_call"]
        self.accepted_total = int(z["accepted_total"])  # includes the W initial placements
        self.total_steps = int(z["total_steps"])
        self.np_t = po.np_dtype(self.dtype)
        if "init_pos" in z:
            self.init_pos, self.init_logp = z["init_pos"], z["init_logp"]
        else:  # digest-only fixture: inputs follow the splitmix64 recipe, pinned by their digests
            self.init_pos = po.init_positions(self.dtype, self.W, self.D, salt=0)
            assert sha(self.init_pos) == str(z["init_pos_sha256"])
            self.init_logp = po.Oracle(self.W, self.D, self.calc, self.params, dtype=self.dtype).logp(self.init_pos)
            assert sha(self.init_logp) == str(z["init_logp_sha256"])
        self.full_steps = sorted(int(k.split("_")[-1]) for k in z.files if k.startswith("chain_step_"))
        self.digest_steps = sorted(int(k.split("_")[-1]) for k in z.files if k.startswith("chain_sha256_step_"))

    def check_chain_step(self, k, step_positions):
        """step_positions: (W, D) positions after stored step k (1-based; 0 = initial placement)."""
        if k in self.full_steps:
            np.testing.assert_array_equal(step_positions, self.z["chain_step_%d" % k],
                                          err_msg="%s: chain step %d differs from the reference" % (self.name, k))
        elif k in self.digest_steps:
            rows = self.z["sample_rows"]
            np.testing.assert_array_equal(step_positions[rows], self.z["chain_rows_step_%d" % k])
            assert sha(step_positions) == str(self.z["chain_sha256_step_%d" % k]), \
                "%s: chain step %d digest differs from the reference" % (self.name, k)
        else:
            raise KeyError(k)

    @property
    def checked_steps(self):
        return self.full_steps + self.digest_steps


def ar_chain(n, W, D, seed, phi, dtype=np.float64):
    """Deterministic [n][W][D] chain for the analysis tests: every (walker, parameter) series is an AR(1) process
    x_t = phi[p] x_{t-1} + e_t whose innovations come from a splitmix64-style integer hash (uniform on [-1, 1)); integer
    arithmetic and one multiply-add per element, so every platform produces the same bytes (pinned by digests in the
    fixtures that use it)."""
    idx = np.arange(n * W * D, dtype=np.uint64).reshape(n, W, D)
    with np.errstate(over="ignore"):
        z = (idx + np.uint64(seed)) * np.uint64(0x9E3779B97F4A7C15)
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        z = z ^ (z >> np.uint64(31))
    e = (z >> np.uint64(11)).astype(np.float64) * (2.0 / 9007199254740992.0) - 1.0
    phi = np.broadcast_to(np.asarray(phi, dtype=np.float64), (D,))
    x = np.empty((n, W, D), dtype=np.float64)
    x[0] = e[0]
    for t in range(1, n):
        x[t] = phi * x[t - 1] + e[t]
    return np.ascontiguousarray(x.astype(dtype))
