"""Bit-faithful restatement of the random streams the reference tests draw their data from
(TEST INFRASTRUCTURE ONLY, like everything under oracle/).

  * StableRNGs.jl 1.0.x `LehmerRNG` (alias `StableRNG`): 128-bit multiplicative congruential generator,
    state ← state · 0x45a31efc5a35d971261fd0407a968add (mod 2^128), output = high 64 bits;
    `StableRNG(seed)`: state = (seed << 1) | 1.                     (benchmarks/Manifest.toml:2539-2543)
  * Julia's `randn(rng::AbstractRNG)` (stdlib Random, normal.jl): 256-layer ziggurat on a 52-bit draw
    (`rand(rng, UInt64) & 0x000fffffffffffff`), sign = lowest bit, layer = next 8 bits, with the tail /
    wedge fall-backs; `rand(rng)` = reinterpret(0x3ff0… | 52 bits) − 1.  The tables ki/wi/fi are regenerated
    with the published recipe (Marsaglia–Tsang / Doornik; r = 3.6541528853610088, area 0.00492867323399,
    2^51 scaling) instead of being copied.
  * Distributions.jl: `rand(rng, Normal(μ, σ)) = μ + σ·randn(rng)`; `rand(rng, MvNormal(μ, Σ)) = μ + chol(Σ).L·randn(d)`.

Nothing here can be checked against a Julia installation (none in the image).  It is pinned the only way
available: data regenerated with it reproduce the free-energy values asserted in the reference's own tests
(test/models/statespace/mlgssm_test.jl:128 → 6275.9015944677, ulgssm_tests.jl:48 → 1854.297647) to all
printed digits — see tests/test_golden_reference.py."""
import math
import struct

import numpy as np

_M128 = (1 << 128) - 1
_MULT = 0x45A31EFC5A35D971261FD0407A968ADD
_R = 3.65415288536100879635194725185604664812733315920964488827246397029393565706474
_AREA = 0.00492867323399


def _ziggurat_tables():
    nm = float(1 << 51)
    ki, wi, fi = [0] * 256, [0.0] * 256, [0.0] * 256
    x1 = _R
    wi[255] = x1 / nm
    fi[255] = math.exp(-0.5 * x1 * x1)
    ki[0] = int(x1 * fi[255] / _AREA * nm)
    wi[0] = _AREA / fi[255] / nm
    fi[0] = 1.0
    for i in range(254, 0, -1):
        x = math.sqrt(-2.0 * math.log(_AREA / x1 + fi[i + 1]))
        ki[i + 1] = int(x / x1 * nm)
        wi[i] = x / nm
        fi[i] = math.exp(-0.5 * x * x)
        x1 = x
    ki[1] = 0
    return ki, wi, fi


_KI, _WI, _FI = _ziggurat_tables()


class StableRNG:
    def __init__(self, seed):
        if seed < 0 or seed >= (1 << 64):
            raise ValueError("seed must fit in UInt64")
        self.state = ((seed << 1) | 1) & _M128

    def rand_u64(self):
        self.state = (self.state * _MULT) & _M128
        return self.state >> 64

    def _rand52(self):
        return self.rand_u64() & 0x000FFFFFFFFFFFFF

    def rand(self):
        """rand(rng) :: Float64 in [0, 1)"""
        return struct.unpack("<d", struct.pack("<Q", 0x3FF0000000000000 | self._rand52()))[0] - 1.0

    def randn(self):
        while True:
            r = self._rand52()
            rabs = r >> 1
            idx = rabs & 0xFF
            x = (-rabs if (r & 1) else rabs) * _WI[idx]
            if rabs < _KI[idx]:
                return x
            if idx == 0:  # tail
                while True:
                    xx = -(1.0 / _R) * math.log(self.rand())
                    yy = -math.log(self.rand())
                    if yy + yy > xx * xx:
                        return (-_R - xx) if ((rabs >> 8) & 1) else (_R + xx)
            elif (_FI[idx - 1] - _FI[idx]) * self.rand() + _FI[idx] < math.exp(-0.5 * x * x):
                return x  # wedge
            # else: draw again

    def normal(self, mu, sigma):
        """rand(rng, Normal(mu, sigma))"""
        return mu + sigma * self.randn()

    def mvnormal(self, mu, cov):
        """rand(rng, MvNormal(mu, cov))"""
        L = np.linalg.cholesky(np.asarray(cov, dtype=np.float64))
        z = np.array([self.randn() for _ in range(len(mu))])
        return np.asarray(mu, dtype=np.float64) + L @ z
