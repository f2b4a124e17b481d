"""The CPU oracle against (1) the reference's own compiled CPU path where it is available,
(2) the committed golden vectors that build produced, (3) analytic two-body answers.
No GPU needed."""
import ctypes

import numpy as np
import pytest

from conftest import bits, load_golden, same_bits


def _run(fn, x0, steps, **kw):
    x, v, a = x0.copy(), np.zeros_like(x0), np.zeros_like(x0)
    fn(x, a, v, steps=steps, **kw)
    return x, v, a


# ---- pinned to the reference ---------------------------------------------------------------

def test_inplace_oracle_matches_golden_reference_outputs(oracle):
    """oracle_step_inplace == the reference's CPU_compute (validation.cpp:28-52), bit for bit,
    on the outputs the reference build produced (tests/golden/ref_cpu_n1024.npz)."""
    g = load_golden("ref_cpu_n1024.npz")
    x, v, a = g["x0"].copy(), np.zeros_like(g["x0"]), np.zeros_like(g["x0"])
    done = 0
    for K in (1, 10, 100):
        oracle.step_inplace(x, a, v, dt=oracle.REF_DT, eps2=oracle.REF_EPS2, steps=K - done)
        done = K
        assert np.array_equal(bits(x), bits(g[f"x_{K}"])), f"positions differ after {K} steps"
        assert np.array_equal(bits(v), bits(g[f"v_{K}"])), f"velocities differ after {K} steps"
        assert np.array_equal(bits(a), bits(g[f"a_{K}"])), f"accelerations differ after {K} steps"


def test_inplace_oracle_matches_reference_at_its_shipped_size(oracle, nb):
    """The reference's own configuration (constants.h:13,25-26: N_BODIES 8192, DT 0.1f, EPS2 0.002f, unseeded
    fill_with_random4): the restatement of validation.cpp:28-52 reproduces the reference build's outputs
    (tests/golden/ref_cpu_n8192.npz) bit for bit after 1 and 10 steps, starting from the product's generator."""
    import ctypes
    g = load_golden("ref_cpu_n8192.npz")
    ctypes.CDLL(None).srand(1)
    x0 = nb.engine.libc_random_bodies(8192)
    assert np.array_equal(bits(x0[:8]), bits(g["x0_head"]))
    x, v, a = x0.copy(), np.zeros_like(x0), np.zeros_like(x0)
    oracle.step_inplace(x, a, v, steps=1)
    assert np.array_equal(bits(x), bits(g["x_1"])) and np.array_equal(bits(a), bits(g["a_1"]))
    oracle.step_inplace(x, a, v, steps=9)
    assert np.array_equal(bits(x), bits(g["x_10"])) and np.array_equal(bits(v), bits(g["v_10"]))
    assert np.array_equal(bits(a), bits(g["a_10"]))


def test_inplace_oracle_matches_the_reference_at_the_benchmark_time_step(oracle):
    """BASELINE configs[1], [2], [4] run dt = 0.01; the reference compiles DT 0.1f in (constants.h:26). The fixture
    tests/golden/ref_cpu_plummer_n1024_dt0.01.npz is the REFERENCE's CPU_compute (validation.cpp:28-52, 43-49) built with
    DT 0.01f as its only change (oracle/ref_dt001.cpp), on the seeded Plummer sphere: oracle_step_inplace(dt=0.01f)
    reproduces it bit for bit after 1, 10 and 100 steps — so the restatement's dt is the reference's DT, not a lookalike."""
    g = load_golden("ref_cpu_plummer_n1024_dt0.01.npz")
    assert np.float32(g["dt"]) == np.float32(0.01) and np.float32(g["eps2"]) == oracle.REF_EPS2
    assert np.array_equal(bits(g["x0"]), bits(load_golden("jacobi_plummer_n1024_dt0.01.npz")["x0"]))    # the same bodies as the Jacobi fixture
    x, v, a = g["x0"].copy(), np.zeros_like(g["x0"]), np.zeros_like(g["x0"])
    done = 0
    for K in (1, 10, 100):
        oracle.step_inplace(x, a, v, dt=np.float32(0.01), eps2=oracle.REF_EPS2, steps=K - done)
        done = K
        assert np.array_equal(bits(x), bits(g[f"x_{K}"])), f"positions differ after {K} steps"
        assert np.array_equal(bits(v), bits(g[f"v_{K}"])), f"velocities differ after {K} steps"
        assert np.array_equal(bits(a), bits(g[f"a_{K}"])), f"accelerations differ after {K} steps"
    # a different dt gives different bits (the fixture does pin the time step)
    x2, v2, a2 = g["x0"].copy(), np.zeros_like(g["x0"]), np.zeros_like(g["x0"])
    oracle.step_inplace(x2, a2, v2, dt=oracle.REF_DT, eps2=oracle.REF_EPS2, steps=1)
    assert not np.array_equal(bits(x2), bits(g["x_1"]))


def test_inplace_oracle_matches_live_reference_build_at_dt001(oracle):
    """The same against the live DT 0.01f build of the reference (oracle/_ref/libref_cpu_dt001.so) on other inputs and sizes."""
    if not oracle.have_ref_dt001():
        pytest.skip("oracle/_ref/libref_cpu_dt001.so not present")
    for n, seed, K in ((1, 1, 3), (2, 2, 5), (37, 3, 7), (513, 4, 4), (1024, 5, 12)):
        rng = np.random.default_rng(seed)
        x0 = rng.normal(0, 1, (n, 4)).astype(np.float32)
        x0[:, 3] = rng.uniform(0.1, 1.0, n).astype(np.float32) / n
        r = _run(oracle.ref_step_dt001, x0, K)
        o = _run(oracle.step_inplace, x0, K, dt=np.float32(0.01), eps2=oracle.REF_EPS2)
        for i in range(3):
            assert np.array_equal(bits(r[i]), bits(o[i])), (n, K, i)


def test_inplace_oracle_matches_the_reference_at_configs1_full_size(oracle, nb):
    """BASELINE configs[1] at FULL size (N = 65536, dt = 0.01, the bench's seeded Plummer sphere): one step of the restatement equals the
    REFERENCE build's CPU_compute (DT 0.01f, tests/golden/ref_cpu_plummer_n65536_dt0.01_sample.npz: 2048 sampled bodies) bit for bit.
    (4.3e9 pair terms on one core: about 9 s. The N = 262144 sample of the same kind is used by the GPU tests.)"""
    g = load_golden("ref_cpu_plummer_n65536_dt0.01_sample.npz")
    n = int(g["n"])
    x0 = nb.engine.seeded_bodies(n, 1, 12345)
    assert np.array_equal(bits(x0[:8]), bits(g["x0_head"]))
    x, v, a = x0.copy(), np.zeros_like(x0), np.zeros_like(x0)
    oracle.set_threads(1)
    oracle.step_inplace(x, a, v, dt=np.float32(0.01), eps2=oracle.REF_EPS2, steps=1)
    idx = g["idx"]
    assert np.array_equal(bits(x[idx]), bits(g["x_1"])) and np.array_equal(bits(v[idx]), bits(g["v_1"]))
    assert np.array_equal(bits(a[idx]), bits(g["a_1"]))


def test_pair_matches_golden_reference_pairs(oracle):
    g = load_golden("ref_pairs.npz")
    for k in range(len(g["bi"])):
        got = oracle.pair(g["bi"][k], g["bj"][k], g["ai"][k])
        assert np.array_equal(bits(got), bits(g["out"][k])), k


def test_inplace_oracle_matches_live_reference_build(oracle):
    """Same check against the live reference library, on different inputs and sizes, where
    oracle/_ref exists (the build container; the prebuilt file also travels to the GPU box)."""
    if not oracle.have_ref():
        pytest.skip("oracle/_ref/libref_cpu.so not present")
    for n, seed, K in ((1, 1, 3), (2, 2, 5), (37, 3, 7), (513, 4, 4), (1024, 5, 12)):
        rng = np.random.default_rng(seed)
        x0 = rng.uniform(-1e5, 1e5, (n, 4)).astype(np.float32)
        x0[:, 3] = rng.uniform(1e5, 1e9, n).astype(np.float32)
        r = _run(oracle.ref_step, x0, K)
        o = _run(oracle.step_inplace, x0, K, dt=oracle.REF_DT, eps2=oracle.REF_EPS2)
        for i in range(3):
            assert np.array_equal(bits(r[i]), bits(o[i])), (n, K, i)


def test_fill_with_random4_known_answer(oracle):
    """utils.cpp:30-37 on glibc from the unseeded state: body 0 is the value SURVEY.md A.2 Q8
    recorded from the reference build, and the whole array equals the golden x0."""
    libc = ctypes.CDLL(None)
    libc.srand(1)
    x = oracle.fill_with_random4_libc(1024)
    assert np.allclose(x[0], [68037.547, -21123.414, 56619.844, 798460160.0], rtol=1e-7)
    assert np.array_equal(bits(x), bits(load_golden("ref_cpu_n1024.npz")["x0"]))
    if oracle.have_ref():
        libc.srand(1)
        assert np.array_equal(bits(oracle.ref_fill_with_random4(1024)), bits(x))


def test_product_fill_with_random4_is_the_reference_generator(oracle, nb):
    libc = ctypes.CDLL(None)
    libc.srand(1)
    x = nb.engine.libc_random_bodies(1024)
    assert np.array_equal(bits(x), bits(load_golden("ref_cpu_n1024.npz")["x0"]))


# ---- internal consistency --------------------------------------------------------------------

def test_jacobi_oracle_matches_its_golden(oracle):
    for name in ("jacobi_refinit_n1024_dt0.1.npz", "jacobi_refinit_n1000_dt0.1.npz", "jacobi_plummer_n1024_dt0.01.npz"):
        g = load_golden(name)
        Ks = sorted(int(k[2:]) for k in g.files if k.startswith("x_"))
        x, v, a = g["x0"].copy(), np.zeros_like(g["x0"]), np.zeros_like(g["x0"])
        done = 0
        for K in Ks:
            oracle.step_jacobi(x, a, v, dt=float(g["dt"]), eps2=float(g["eps2"]), steps=K - done)
            done = K
            assert np.array_equal(bits(x), bits(g[f"x_{K}"])), (name, K)
            assert np.array_equal(bits(v), bits(g[f"v_{K}"])), (name, K)
            assert np.array_equal(bits(a), bits(g[f"a_{K}"])), (name, K)


def test_blocked_accel_equals_scalar_pair_loop(oracle):
    """The SIMD-blocked Jacobi kernel is the plain sequential sum of oracle_pair terms."""
    rng = np.random.default_rng(11)
    n = 150
    x = rng.uniform(-100, 100, (n, 4)).astype(np.float32)
    x[:, 3] = rng.uniform(1, 1e3, n).astype(np.float32)
    for (i0, i1, j0, j1) in ((0, n, 0, n), (5, 77, 0, n), (10, 140, 30, 90), (0, 16, 16, 32), (3, 4, 0, n)):
        got = oracle.accel_range(x, i0, i1, j0, j1, eps2=0.01)
        want = np.zeros_like(got)
        for i in range(i0, i1):
            acc = np.zeros(4, np.float32)
            for j in range(j0, j1):
                if j != i:
                    acc = oracle.pair(x[i], x[j], acc, eps2=0.01)
            want[i - i0] = acc
        assert np.array_equal(bits(got), bits(want)), (i0, i1, j0, j1)


def test_jacobi_first_body_equals_inplace_first_body(oracle):
    """Body 0 sees only un-advanced positions in the reference's in-place order too, so both
    orders must agree on it exactly; later bodies differ only through already-moved sources."""
    g = load_golden("ref_cpu_n1024.npz")
    xi, vi, ai = _run(oracle.step_inplace, g["x0"], 1, dt=oracle.REF_DT, eps2=oracle.REF_EPS2)
    xj, vj, aj = _run(oracle.step_jacobi, g["x0"], 1, dt=oracle.REF_DT, eps2=oracle.REF_EPS2)
    assert np.array_equal(bits(xi[0]), bits(xj[0])) and np.array_equal(bits(ai[0]), bits(aj[0]))
    # in-place vs Jacobi after one step, reference init: <= 1e-6 of the 1e5 scale
    assert np.abs(xi - xj)[:, :3].max() / 1e5 < 1e-6


# ---- analytic known answers --------------------------------------------------------------------

def test_two_body_analytic(oracle):
    """a = m r / (r^2 + eps2)^(3/2) along the separation; integrator is half-kick + drift
    (validation.cpp:43-49): v1 = 0.5*dt*a, x1 = x0 + dt*v1."""
    eps2, dt = 0.002, 0.1
    x = np.array([[0, 0, 0, 5.0], [3, 4, 0, 7.0]], np.float32)
    v, a = np.zeros_like(x), np.zeros_like(x)
    oracle.step_inplace(x.copy(), a, v.copy(), dt=dt, eps2=eps2)  # accelerations of body 0 are un-advanced
    d = 25.0 + eps2
    a0 = 7.0 / d ** 1.5 * np.array([3, 4, 0.0])
    assert np.allclose(a[0, :3], a0, rtol=2e-6)
    xs, vs, as_ = _run(oracle.step_jacobi, x, 1, dt=dt, eps2=eps2)
    a1 = -5.0 / d ** 1.5 * np.array([3, 4, 0.0])
    assert np.allclose(as_[1, :3], a1, rtol=2e-6)
    assert np.allclose(vs[0, :3], 0.5 * dt * a0, rtol=2e-6)
    assert np.allclose(xs[0, :3], dt * 0.5 * dt * a0, rtol=2e-6)
    assert xs[0, 3] == 5.0 and xs[1, 3] == 7.0          # mass untouched
    assert as_[0, 3] == 0.0 and vs[0, 3] == 0.0          # w lanes stay 0


def test_overflow_corner_is_the_references(oracle):
    """Q13: d*d*d overflows fp32 for r >~ 2.6e6, so 1/sqrtf(inf) = 0 and the pair adds nothing."""
    far = oracle.pair([0, 0, 0, 1.0], [3e6, 0, 0, 1e9], [0, 0, 0, 0])
    assert np.all(far == 0)
    near = oracle.pair([0, 0, 0, 1.0], [2e6, 0, 0, 1e9], [0, 0, 0, 0])
    assert near[0] > 0


def test_verify_still_bodies_counts(oracle, nb):
    """validation.cpp:143-164: 1 % of the smaller magnitude, per component, w ignored."""
    x = np.array([[100, 100, 100, 1], [100, 100, 100, 1], [0, 5, 5, 1], [-100, 100, 100, 9]], np.float32)
    v = np.array([[100.9, 100, 100, 2], [101.2, 100, 100, 1], [1e-9, 5, 5, 1], [-100.5, 99.5, 100, 1]], np.float32)
    assert oracle.verify_still_bodies(v, x) == 2           # bodies 1 (1.2 %) and 2 (zero-magnitude tolerance)
    assert nb.engine.verify_still_bodies(v, x) == 2
    assert oracle.verify_still_bodies(x, x) == 0
    assert oracle.verify_equality4(v, x) == 3 and nb.engine.verify_equality4(v, x) == 3  # body 2 is within 0.01
    rng = np.random.default_rng(5)
    p = rng.normal(0, 1, (500, 4)).astype(np.float32)
    q = (p * (1 + rng.normal(0, 0.008, p.shape))).astype(np.float32)
    assert oracle.verify_still_bodies(q, p) == nb.engine.verify_still_bodies(q, p)
    assert oracle.verify_equality4(q, p) == nb.engine.verify_equality4(q, p)


def test_f64acc_oracle_is_close_to_fp32_oracle(oracle):
    g = load_golden("jacobi_plummer_n1024_dt0.01.npz")
    a32 = oracle.accel_range(g["x0"], 0, 1024, eps2=0.002)
    a64 = oracle.accel_range(g["x0"], 0, 1024, eps2=0.002, f64acc=True)
    scale = np.abs(a64[:, :3]).max()
    assert np.abs(a32 - a64)[:, :3].max() / scale < 5e-6


def test_oracle_matches_live_reference_on_hostile_inputs(oracle):
    """Randomised: magnitudes from 1e-18 to 1e18 (d*d*d overflow and underflow), coincident bodies,
    zero and negative masses, non-zero initial velocities — the restatement and the reference's own
    CPU_compute must agree bit for bit, NaN for NaN."""
    if not oracle.have_ref():
        pytest.skip("oracle/_ref/libref_cpu.so not present")
    rng = np.random.default_rng(2718)
    for case in range(60):
        n = int(rng.integers(1, 48))
        mag = 10.0 ** rng.uniform(-18, 18)
        x0 = (rng.normal(0, 1, (n, 4)) * mag).astype(np.float32)
        x0[:, 3] = (rng.normal(0, 1, n) * 10.0 ** rng.uniform(-10, 12)).astype(np.float32)
        if n > 3:
            x0[1, :3] = x0[0, :3]                      # coincident pair
            x0[2, 3] = 0.0                             # massless body
        v0 = (rng.normal(0, 1, (n, 4)) * mag * 0.1).astype(np.float32)
        v0[:, 3] = 0
        outs = []
        for fn, kw in ((oracle.ref_step, {}), (oracle.step_inplace, dict(dt=oracle.REF_DT, eps2=oracle.REF_EPS2))):
            x, v, a = x0.copy(), v0.copy(), np.zeros_like(x0)
            with np.errstate(all="ignore"):
                fn(x, a, v, steps=3, **kw)
            outs.append((x, v, a))
        for p, q in zip(*outs):
            assert np.array_equal(bits(p), bits(q)), (case, n, mag)
