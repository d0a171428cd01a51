# dump_rxinfer_reference.jl — to be run by someone WITH a Julia toolchain, from the RxInfer.jl checkout:
#
#     julia --project=. path/to/tests/golden/dump_rxinfer_reference.jl path/to/tests/golden
#
# Writes rxinfer_reference.json next to the other fixtures.  tests/test_golden_reference.py picks the file up when it exists
# and turns three things that this repository can only bound or assume into hard checks (DESIGN.md §5, ADVICE r1):
#   (1) the label vector and observations of test/models/mixtures/gmm_univariate_tests.jl (the golden 284.76 is not
#       reproducible from `rand(StableRNG(12345), Categorical([1/3, 2/3]), 150)` as restated here);
#   (2) the per-iteration posteriors and free energies of that model (gates the ASSUMED mean-field update order of the
#       mixture engines: q(z) from the previous marginals, then q(s), q(m), then q(p) with the new q(m));
#   (3) the first UInt64 / rand / randn draws of StableRNG(12345) and the first Categorical draws (pins oracle/stable_rng.py
#       directly instead of through free-energy values).
using RxInfer, StableRNGs, Distributions, Random

outdir = length(ARGS) > 0 ? ARGS[1] : @__DIR__

@model function univariate_gaussian_mixture_model(y)
    s ~ Beta(1.0, 1.0)
    m[1] ~ Normal(mean = -2.0, variance = 1e3)
    p[1] ~ Gamma(shape = 0.01, rate = 0.01)
    m[2] ~ Normal(mean = 2.0, variance = 1e3)
    p[2] ~ Gamma(shape = 0.01, rate = 0.01)
    for i in eachindex(y)
        z[i] ~ Bernoulli(s)
        y[i] ~ NormalMixture(switch = z[i], m = m, p = p)
    end
end

init = @initialization begin
    q(s) = vague(Beta)
    q(m) = [NormalMeanVariance(-2.0, 1e3), NormalMeanVariance(2.0, 1e3)]
    q(p) = [vague(GammaShapeRate), vague(GammaShapeRate)]
end

rng = StableRNG(12345)
n = 150
z = rand(rng, Categorical([1 / 3, 2 / 3]), n)
dists = [Normal(-10.0, sqrt(inv(3.777))), Normal(10.0, sqrt(inv(0.333)))]
y = [rand(rng, dists[z[i]]) for i in 1:n]

result = infer(model = univariate_gaussian_mixture_model(), data = (y = y,), constraints = MeanField(), returnvars = KeepEach(),
               free_energy = Float64, iterations = 10, initialization = init)

r2 = StableRNG(12345)
draws = (u64 = [rand(r2, UInt64) for _ in 1:4], rand = [rand(r2) for _ in 1:4], randn = [randn(r2) for _ in 1:4],
         cat = rand(StableRNG(7), Categorical([0.1, 0.2, 0.3, 0.25, 0.15]), 32))

jl(x::AbstractVector) = "[" * join(jl.(x), ",") * "]"
jl(x::Real) = repr(Float64(x))
jl(x::Integer) = string(x)
open(joinpath(outdir, "rxinfer_reference.json"), "w") do io
    print(io, "{\"z\":", jl(z), ",\"y\":", jl(y), ",\"free_energy\":", jl(result.free_energy))
    for (name, f) in (("m_mean", q -> mean(q)), ("m_var", q -> var(q)), ("p_shape", q -> shape(q)), ("p_rate", q -> rate(q)))
        v = name[1] == 'm' ? result.posteriors[:m] : result.posteriors[:p]
        print(io, ",\"", name, "\":[", join((jl([f(v[it][k]) for k in 1:2]) for it in 1:10), ","), "]")
    end
    print(io, ",\"s_params\":[", join((jl(collect(params(result.posteriors[:s][it]))) for it in 1:10), ","), "]")
    print(io, ",\"draws\":{\"u64\":[", join(string.(draws.u64), ","), "],\"rand\":", jl(draws.rand), ",\"randn\":", jl(draws.randn),
          ",\"categorical\":", jl(draws.cat), "}}")
end
println("wrote ", joinpath(outdir, "rxinfer_reference.json"))
