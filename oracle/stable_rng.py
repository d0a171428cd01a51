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

    def rand_range(self, n):
        """rand(rng, 1:n) for a StableRNG: StableRNGs pins `SamplerRangeFast` for its generator — the low
        bits of one UInt64 under the mask of n − 1, rejection above n − 1.  Returns a 0-based index."""
        m = n - 1
        mask = (1 << m.bit_length()) - 1
        while True:
            x = self.rand_u64() & mask
            if x <= m:
                return x

    def categorical_alias_table(self, probs):
        """`rand(rng, Categorical(p))` through AliasTables.jl 1.1 (Distributions ≥ 0.25.109; benchmarks/Manifest.toml:158-162):
        one UInt64 per draw, the top bits select a cell of the power-of-two table, the remaining bits, left-aligned, are
        compared with the cell's redirect threshold (values below it go to the cell's alias).  A deficient cell draws from the
        first element that still has surplus; a donor that falls short in its own cell joins the queue.  Layout restated from the package's
        documentation, not from its source (absent here) — see tests/test_golden_reference.py for what pins it.  0-based."""
        table = getattr(self, "_alias_cache", {}).get(tuple(probs))
        if table is None:
            K = len(probs)
            n = 1 << max(K - 1, 0).bit_length() if K > 1 else 1
            cap = 1.0 / n
            rem = [float(p) / sum(probs) for p in probs] + [0.0] * (n - K)   # mass of element i still to be placed
            table = [(0.0, i) for i in range(n)]                              # (fraction of the cell that redirects, where to)
            queue = [i for i in range(n) if rem[i] < cap - 1e-18]             # deficient cells, in index order
            while queue:
                i = queue.pop(0)
                need = cap - rem[i]
                donor = next((j for j in range(n) if rem[j] > cap + 1e-18), None)
                if donor is None:
                    break                                                      # rounding residue only
                table[i] = (need / cap, donor)
                rem[donor] -= need
                if rem[donor] < cap - 1e-18:
                    queue.append(donor)                                        # the donor's own cell is now short
            self._alias_cache = getattr(self, "_alias_cache", {})
            self._alias_cache[tuple(probs)] = table
        n = len(table)
        shift = (n - 1).bit_length()
        x = self.rand_u64()
        cell = x >> (64 - shift) if shift else 0
        val = (x << shift) & ((1 << 64) - 1)
        frac, alias = table[cell]
        return alias if val < int(frac * 2.0 ** 64) else cell

    def categorical_legacy_alias(self, probs):
        """The sampler Distributions used before AliasTables.jl (StatsBase.make_alias_table! + two draws per sample:
        `i = rand(rng, 1:K)`, `u = rand(rng)`, `u < accept[i] ? i : alias[i]`).  0-based."""
        K = len(probs)
        key = ("legacy",) + tuple(probs)
        tab = getattr(self, "_alias_cache", {}).get(key)
        if tab is None:
            a = [p * K / sum(probs) for p in probs]
            alias = list(range(K))
            larges = [i for i in range(K) if a[i] > 1.0]
            smalls = [i for i in range(K) if a[i] < 1.0]
            while larges and smalls:
                sm, lg = smalls.pop(), larges.pop()
                alias[sm] = lg
                a[lg] = (a[lg] - 1.0) + a[sm]
                (larges if a[lg] > 1.0 else smalls).append(lg)
            for i in smalls:
                a[i] = 1.0
            tab = (a, alias)
            self._alias_cache = getattr(self, "_alias_cache", {})
            self._alias_cache[key] = tab
        a, alias = tab
        i = self.rand_range(K)
        return i if self.rand() < a[i] else alias[i]

    def categorical_inverse_cdf(self, probs):
        """`rand(rng, d::DiscreteNonParametric)` for a SINGLE draw: inverse CDF on one `rand(rng)`.  0-based."""
        u, c = self.rand(), 0.0
        for i, p in enumerate(probs):
            c += p
            if u < c:
                return i
        return len(probs) - 1

    def mvnormal(self, mu, cov):
        """rand(rng, MvNormal(mu, cov))"""
        L = np.linalg.cholesky(np.asarray(cov, dtype=np.float64))
        z = np.array([self.randn() for _ in range(len(mu))])
        return np.asarray(mu, dtype=np.float64) + L @ z
