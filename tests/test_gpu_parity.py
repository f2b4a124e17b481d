"""Parity of the HIP path (through the C-ABI) against the CPU oracle and the committed goldens.

Tiers:
  T1  strict kernel == Jacobi oracle, bit for bit (x, v, a), any N, any K.
  T2  fast kernel, one step from identical state: positions rel-err <= 1e-6 of the system scale,
      accelerations <= 1e-5 of max|a|.
  T3  fast kernel trajectory, Plummer N=1024 dt=0.01 K=100: max|dx| <= 1e-5 (scale radius 1).
  T4  the reference's own 1 % rule (validation.cpp:143-164) vs the literal in-place reference
      outputs: reported and bounded.
"""
import numpy as np
import pytest
import torch

from conftest import bits, load_golden, same_bits

pytestmark = pytest.mark.gpu


def _gpu_run(nb, x0, steps, dt, eps2, kernel, **opts):
    sim = nb.engine.Simulation(x0, dt=dt, eps2=eps2, kernel=kernel, **opts)
    sim.run(steps)
    return sim.state()


def _rand_bodies(n, seed, scale=1e5, mlo=1e5, mhi=1e9):
    rng = np.random.default_rng(seed)
    x = rng.uniform(-scale, scale, (n, 4)).astype(np.float32)
    x[:, 3] = rng.uniform(mlo, mhi, n).astype(np.float32)
    return x


# ---- T1: strict == oracle, bitwise -----------------------------------------------------------

@pytest.mark.parametrize("name", ["jacobi_refinit_n1024_dt0.1.npz", "jacobi_refinit_n1000_dt0.1.npz",
                                  "jacobi_plummer_n1024_dt0.01.npz"])
def test_strict_matches_golden_bitwise(nb, name):
    g = load_golden(name)
    Ks = sorted(int(k[2:]) for k in g.files if k.startswith("x_"))
    for K in Ks:
        x, v, a = _gpu_run(nb, g["x0"], K, float(g["dt"]), float(g["eps2"]), nb.KERNEL_STRICT)
        assert same_bits(x, g[f"x_{K}"]), (name, K, "x")
        assert same_bits(v, g[f"v_{K}"]), (name, K, "v")
        assert same_bits(a, g[f"a_{K}"]), (name, K, "a")


@pytest.mark.parametrize("n,K", [(1, 2), (2, 3), (37, 4), (255, 2), (256, 2), (257, 2), (1025, 2), (4096, 2), (5000, 1)])
def test_strict_matches_oracle_bitwise_ragged(nb, oracle, n, K):
    x0 = _rand_bodies(n, 100 + n)
    xo, vo, ao = x0.copy(), np.zeros_like(x0), np.zeros_like(x0)
    oracle.step_jacobi(xo, ao, vo, dt=0.1, eps2=0.002, steps=K)
    x, v, a = _gpu_run(nb, x0, K, 0.1, 0.002, nb.KERNEL_STRICT)
    assert same_bits(x, xo) and same_bits(v, vo) and same_bits(a, ao)


def test_strict_overflow_corner_and_coincident_bodies(nb, oracle):
    """r > 2.6e6 pairs contribute exactly 0 (d^3 overflows, Q13); coincident bodies add 0."""
    x0 = _rand_bodies(300, 5)
    x0[10, :3] = [4e6, 0, 0]
    x0[11, :3] = x0[12, :3]
    xo, vo, ao = x0.copy(), np.zeros_like(x0), np.zeros_like(x0)
    oracle.step_jacobi(xo, ao, vo, dt=0.1, eps2=0.002)
    x, v, a = _gpu_run(nb, x0, 1, 0.1, 0.002, nb.KERNEL_STRICT)
    assert same_bits(a, ao) and same_bits(x, xo)
    assert np.all(a[10, :3] == 0)


@pytest.mark.parametrize("kernel,shape", [("fast", None), ("onesided", None), ("symmetric", (1, 2)), ("symmetric", (2, 4))])
def test_fast_kernels_at_the_overflow_corner(nb, kernel, shape):
    """What the FAST / ONESIDED / SYMMETRIC kernels do where the reference's `d*d*d` overflows (r > 2.6e6: the reference's
    term is exactly 0, validation.cpp:16 == kernel.cu:20). They evaluate rsq(d)^3, which does NOT
    overflow there, so they keep the physically correct (tiny) term m/r^2 the reference drops. Stated, bounded and finite:
      * every value finite, also for the far body (whose every pair is beyond the overflow radius);
      * against the fp64 sum WITHOUT the overflow: the usual 1e-5 of max|a|; the far body itself to 1e-4 of its own |a|;
      * against the strict kernel (the reference's arithmetic): the difference is that one dropped term, at most
        m_far / r_min^2, plus the usual tolerance.
    DESIGN.md 2 and include/nbody.h state the same."""
    n, far = 1500, 10
    x0 = _rand_bodies(n, 5)
    x0[far, :3] = [4e6, 0, 0]
    x0[far, 3] = 1e9
    k = {"fast": nb.KERNEL_FAST, "onesided": nb.KERNEL_ONESIDED, "symmetric": nb.KERNEL_SYMMETRIC}[kernel]
    sim = nb.engine.Simulation(x0, dt=0.1, eps2=0.002, kernel=k)
    if shape:
        sim.ctx.set_symmetric_shape(*shape)
        assert sim.ctx.step_info(n)["symmetric"]
    sim.run(1)
    x, v, a = sim.state()
    xs, vs, a_strict = _gpu_run(nb, x0, 1, 0.1, 0.002, nb.KERNEL_STRICT)
    assert np.all(np.isfinite(a)) and np.all(np.isfinite(x)) and np.all(np.isfinite(v))
    assert np.all(a_strict[far, :3] == 0)                                   # the reference: every pair of the far body overflows
    p = x0[:, :3].astype(np.float64)
    r = p[None, :, :] - p[:, None, :]
    d = (r * r).sum(-1) + np.float64(np.float32(0.002))
    w = x0[None, :, 3].astype(np.float64) * d ** -1.5
    w[np.arange(n), np.arange(n)] = 0.0
    truth = (r * w[:, :, None]).sum(1)
    amax = np.abs(truth).max()
    others = np.arange(n) != far
    assert np.abs(a[others, :3] - truth[others]).max() <= 1e-5 * amax
    assert np.abs(a[far, :3] - truth[far]).max() <= 1e-4 * np.abs(truth[far]).max()
    assert np.abs(truth[far]).max() > 0
    dropped = 1e9 / ((4e6 - 1e5) ** 2)                                       # the largest term the reference drops
    assert np.abs(a[others, :3] - a_strict[others, :3]).max() <= 1.01 * dropped + 1e-5 * amax


# ---- T2: fast kernel, single step --------------------------------------------------------------

@pytest.mark.parametrize("n,init,dt,scale", [(1024, 0, 0.1, 1e5), (1000, 0, 0.1, 1e5), (1024, 1, 0.01, 1.0),
                                             (4099, 0, 0.1, 1e5), (16384, 1, 0.01, 1.0), (300, 0, 0.1, 1e5)])
def test_fast_single_step(nb, oracle, n, init, dt, scale):
    x0 = nb.engine.seeded_bodies(n, init, 99)
    xo, vo, ao = x0.copy(), np.zeros_like(x0), np.zeros_like(x0)
    oracle.step_jacobi(xo, ao, vo, dt=dt, eps2=0.002)
    x, v, a = _gpu_run(nb, x0, 1, dt, 0.002, nb.KERNEL_FAST)
    amax = np.abs(ao[:, :3]).max()
    assert np.abs(a - ao)[:, :3].max() / amax <= 1e-5
    assert np.abs(x - xo)[:, :3].max() / scale <= 1e-6
    assert np.array_equal(x[:, 3], x0[:, 3])                       # mass carried through
    assert np.all(a[:, 3] == 0) and np.all(v[:, 3] == 0)
    # and as close to the fp64-accumulated truth as the fp32 oracle is (a few ulp of max|a|)
    at = oracle.accel_range(x0, 0, n, eps2=0.002, f64acc=True)
    assert np.abs(a - at)[:, :3].max() / amax <= 1e-5


def test_fast_single_step_vs_literal_reference(nb, oracle):
    """GPU (Jacobi) vs the reference's in-place CPU_compute outputs after ONE step from the
    reference's own initial conditions (golden from the reference build). Positions agree to
    1e-6 of the scale and pass the reference's 1 % rule. Accelerations of bodies whose close
    neighbours were already advanced by the in-place loop differ by that ordering effect (an
    oracle-vs-oracle quantity, computed here), never by more than it plus 1e-5."""
    g = load_golden("ref_cpu_n1024.npz")
    x, v, a = _gpu_run(nb, g["x0"], 1, 0.1, 0.002, nb.KERNEL_FAST)
    assert np.abs(x - g["x_1"])[:, :3].max() / 1e5 <= 1e-6
    assert nb.engine.verify_still_bodies(x, g["x_1"]) == 0         # the reference's own 1 % rule
    aj = oracle.accel_range(g["x0"], 0, 1024, eps2=0.002)          # Jacobi order, same arithmetic
    amax = np.abs(g["a_1"][:, :3]).max()
    ordering = np.abs(aj - g["a_1"])[:, :3]                        # in-place vs Jacobi, per component
    assert np.all(np.abs(a - g["a_1"])[:, :3] <= ordering + 1e-5 * amax)
    # body 0 is processed first by the in-place loop, so it sees no ordering effect at all
    assert np.all(ordering[0] == 0) and np.abs(a[0] - g["a_1"][0])[:3].max() <= 1e-5 * amax


@pytest.mark.parametrize("opts", [dict(tile=256, bodies_per_lane=1, jsplit=1), dict(tile=512, bodies_per_lane=2, jsplit=2),
                                  dict(tile=1024, bodies_per_lane=4, jsplit=3), dict(tile=2048, bodies_per_lane=4, jsplit=1),
                                  dict(tile=256, bodies_per_lane=4, jsplit=32), dict(tile=1024, bodies_per_lane=2, jsplit=7)])
def test_fast_kernel_configurations_agree(nb, oracle, opts):
    n = 3001
    x0 = nb.engine.seeded_bodies(n, 1, 5)
    ao = oracle.accel_range(x0, 0, n, eps2=0.002)
    x, v, a = _gpu_run(nb, x0, 1, 0.01, 0.002, nb.KERNEL_FAST, **opts)
    assert np.abs(a - ao)[:, :3].max() / np.abs(ao[:, :3]).max() <= 1e-5


# ---- the symmetric kernel (every unordered pair once) -----------------------------------------------

@pytest.mark.parametrize("waves,bpl", [(1, 2), (1, 4), (2, 4), (1, 8), (1, 10), (2, 8), (2, 10), (4, 8), (4, 10)])
@pytest.mark.parametrize("n,init", [(1000, 0), (4099, 1), (6144, 0)])
def test_symmetric_kernel_shapes_vs_oracle(nb, oracle, waves, bpl, n, init):
    """Every block shape the library builds, on sizes that are not multiples of the block (the last
    block is padded with massless bodies) and on one that is: accelerations against the fp64-accumulated
    CPU sum, momentum balance, w = 0, and agreement with the one-sided kernel."""
    block = 64 * waves * bpl
    x0 = nb.engine.seeded_bodies(n, init, 7)
    ctx = nb.engine.Context(kernel=nb.KERNEL_SYMMETRIC)
    ctx.set_symmetric_shape(waves, bpl)
    info = ctx.step_info(n)
    nblk = -(-n // block)
    if nblk >= 2:
        assert info["symmetric"] and info["block_bodies"] == block and info["slabs"] == nblk
        assert info["workgroups"] == nblk * (nblk + 1) // 2
        assert info["evaluated_pairs"] == info["workgroups"] * block * block
    else:
        assert not info["symmetric"]                 # a single block: nothing to pair up, one-sided kernel
    x = torch.from_numpy(x0).cuda()
    a = torch.full((n, 4), 3.0, device="cuda")
    ctx.accel_range(x, a, 0, n, 0, n)
    ctx.sync()
    ag = a.cpu().numpy()
    truth = oracle.accel_range(x0, 0, n, eps2=0.002, f64acc=True)
    amax = np.abs(truth[:, :3]).max()
    assert np.abs(ag - truth)[:, :3].max() / amax <= 1e-5
    assert np.all(ag[:, 3] == 0)
    m = x0[:, 3:4].astype(np.float64)
    assert np.abs((m * ag[:, :3]).sum(0)).max() / (m * np.abs(ag[:, :3])).sum() < 1e-6
    one = nb.engine.Context(kernel=nb.KERNEL_ONESIDED)
    a1 = torch.zeros_like(a)
    one.accel_range(x, a1, 0, n, 0, n)
    one.sync()
    assert np.abs(ag - a1.cpu().numpy())[:, :3].max() / amax <= 1e-5
    # accumulate continues a sum already in the output
    ctx.accel_range(x, a, 0, n, 0, n, accumulate=True)
    ctx.sync()
    assert np.abs(a.cpu().numpy() - 2 * ag)[:, :3].max() / amax <= 1e-6


@pytest.mark.parametrize("bpl", [8, 10])
@pytest.mark.parametrize("n,init", [(700, 0), (4099, 1), (6400, 0), (20000, 1)])
def test_symmetric_kernel_in_runs_vs_oracle(nb, oracle, bpl, n, init):
    """The run-based decomposition (independent waves, units of one 64-body chunk): accelerations against the
    fp64-accumulated CPU sums at sizes that are and are not multiples of the 64*bpl I-block, momentum balance, w = 0,
    whole steps equal to force + integrate, run-to-run bitwise reproducibility."""
    x0 = nb.engine.seeded_bodies(n, init, 11)
    ctx = nb.engine.Context(kernel=nb.KERNEL_SYMMETRIC, dt=0.01)
    ctx.set_symmetric_shape(0, bpl)
    ctx.set_symmetric_runs(1)
    info = ctx.step_info(n)
    assert info["symmetric"] and info["runs"] and info["block_bodies"] == 64 * bpl
    x = torch.from_numpy(x0).cuda()
    a = torch.full((n, 4), 3.0, device="cuda")
    ctx.accel_range(x, a, 0, n, 0, n)
    ctx.sync()
    ag = a.cpu().numpy()
    truth = oracle.accel_range(x0, 0, n, eps2=0.002, f64acc=True)
    amax = np.abs(truth[:, :3]).max()
    assert np.abs(ag - truth)[:, :3].max() / amax <= 1e-5
    assert np.all(ag[:, 3] == 0)
    m = x0[:, 3:4].astype(np.float64)
    assert np.abs((m * ag[:, :3]).sum(0)).max() / (m * np.abs(ag[:, :3])).sum() < 1e-6
    a2 = torch.zeros_like(a)
    ctx.accel_range(x, a2, 0, n, 0, n)
    ctx.sync()
    assert torch.equal(a, a2)
    # a whole step through nbody_step: the integrate sums the same slabs
    v = torch.zeros_like(x)
    a3 = torch.zeros_like(x)
    x1 = x.clone()
    ctx.step(x1, a3, v, 1)
    ctx.sync()
    assert torch.equal(a3, a)
    xs, vs = x0.copy(), np.zeros_like(x0)
    oracle.integrate(xs, vs, ag, dt=0.01)
    assert same_bits(x1.cpu().numpy(), xs) and same_bits(v.cpu().numpy(), vs)


def test_symmetric_step_trajectory_and_determinism(nb, oracle):
    """Whole steps through nbody_step with the symmetric kernel forced at a size the oracle finishes:
    K=10 steps vs the Jacobi oracle, bitwise run-to-run reproducibility, integrate bit-exact."""
    g = load_golden("jacobi_plummer_n1024_dt0.01.npz")
    runs = []
    for _ in range(2):
        sim = nb.engine.Simulation(g["x0"], dt=0.01, eps2=0.002, kernel=nb.KERNEL_SYMMETRIC)
        sim.ctx.set_symmetric_shape(1, 2)
        assert sim.ctx.step_info(1024)["symmetric"]
        sim.run(9)
        x9, v9, _ = sim.state()
        sim.run(1)
        runs.append(sim.state())
    x, v, a = runs[0]
    assert np.abs(x - g["x_10"])[:, :3].max() <= 2e-6
    assert all(np.array_equal(p, q) for p, q in zip(runs[0], runs[1]))
    xs, vs = x9.copy(), v9.copy()
    oracle.integrate(xs, vs, a, dt=0.01)
    assert same_bits(xs, x) and same_bits(vs, v)


def test_symmetric_vs_onesided_at_n65536(nb, oracle):
    """configs[1]'s size: FAST picks the symmetric kernel; it agrees with the one-sided kernel to a few
    ulp of max|a| and is at least as close to the fp64 truth on sampled targets."""
    n = 65536
    x0 = nb.engine.seeded_bodies(n, 0, 515)        # the reference's cube and mass range
    x = torch.from_numpy(x0).cuda()
    out = {}
    for name, k in (("fast", nb.KERNEL_FAST), ("one", nb.KERNEL_ONESIDED)):
        ctx = nb.engine.Context(kernel=k)
        assert ctx.step_info(n)["symmetric"] == (name == "fast")
        a = torch.zeros_like(x)
        ctx.accel_range(x, a, 0, n, 0, n)
        ctx.sync()
        out[name] = a.cpu().numpy()
    truth = oracle.accel_range(x0, 5000, 5512, 0, n, eps2=0.002, f64acc=True)
    amax = np.abs(truth[:, :3]).max()
    e_sym = np.abs(out["fast"][5000:5512] - truth)[:, :3].max() / amax
    e_one = np.abs(out["one"][5000:5512] - truth)[:, :3].max() / amax
    assert e_sym <= 1e-5 and e_one <= 1e-5
    an = np.abs(out["one"][:, :3]).max()
    assert np.abs(out["fast"] - out["one"])[:, :3].max() / an <= 2e-5


@pytest.mark.parametrize("waves,bpl", [(0, 0), (1, 2), (2, 8)])
@pytest.mark.parametrize("n,i0,i1,j0,count", [(3000, 1000, 1750, 1750, 1100),   # plain run after the targets
                                              (3000, 2250, 3000, 0, 900),        # run starts at 0
                                              (3000, 1000, 1750, 2500, 1300),    # run wraps past the end
                                              (5000, 100, 133, 4000, 1050),      # few targets, wrapping run
                                              (4096, 0, 2048, 2048, 2048)])      # whole blocks
def test_accel_cross_two_disjoint_sets(nb, oracle, waves, bpl, n, i0, i1, j0, count):
    """nbody_accel_cross: every (target, source) pair once, both sides' accelerations. Targets' sums and sources'
    sums against the fp64-accumulated CPU sums over exactly those pairs; sum of m*a over both sets is zero."""
    x0 = nb.engine.seeded_bodies(n, 0, 77)
    ctx = nb.engine.Context()
    ctx.set_symmetric_shape(waves, bpl)
    x = torch.from_numpy(x0).cuda()
    ai = torch.full((i1 - i0, 4), 5.0, device="cuda")
    aj = torch.full((count, 4), 9.0, device="cuda")
    ctx.accel_cross(x, ai, i0, i1, False, j0, count, aj)
    ctx.sync()
    run = (np.arange(count) + j0) % n
    # reorder so that the run is contiguous: targets first, then the run
    xs = np.ascontiguousarray(np.concatenate([x0[i0:i1], x0[run]]))
    ni = i1 - i0
    want_i = oracle.accel_range(xs, 0, ni, ni, ni + count, eps2=0.002, f64acc=True)
    want_j = oracle.accel_range(xs, ni, ni + count, 0, ni, eps2=0.002, f64acc=True)
    gi, gj = ai.cpu().numpy(), aj.cpu().numpy()
    assert np.abs(gi - want_i)[:, :3].max() / np.abs(want_i[:, :3]).max() <= 1e-5
    assert np.abs(gj - want_j)[:, :3].max() / np.abs(want_j[:, :3]).max() <= 1e-5
    assert np.all(gi[:, 3] == 0) and np.all(gj[:, 3] == 0)
    mi, mj = xs[:ni, 3:4].astype(np.float64), xs[ni:, 3:4].astype(np.float64)
    net = (mi * gi[:, :3]).sum(0) + (mj * gj[:, :3]).sum(0)
    assert np.abs(net).max() / ((mi * np.abs(gi[:, :3])).sum() + (mj * np.abs(gj[:, :3])).sum()) < 1e-6
    # accumulate_i continues the targets' sums; the sources' output is always overwritten
    ctx.accel_cross(x, ai, i0, i1, True, j0, count, aj)
    ctx.sync()
    assert np.abs(ai.cpu().numpy() - 2 * gi)[:, :3].max() / np.abs(gi[:, :3]).max() <= 1e-6
    assert np.array_equal(aj.cpu().numpy(), gj)


def test_randomised_symmetric_and_cross_cases(nb, oracle):
    """30 random cases — size, block shape, length scale (1e-2 ... 1e5), mass range, zero and negative masses,
    coincident bodies: the symmetric kernel over the whole set, and nbody_accel_cross over a random split of it, against
    the fp64-accumulated CPU sums; the two sides of every cross call balance (sum of m*a over both sets = 0)."""
    rng = np.random.default_rng(2718)
    shapes = [(1, 2), (1, 4), (2, 4), (1, 8), (1, 10), (2, 8), (2, 10), (4, 8), (4, 10)]
    for case in range(30):
        w, b = shapes[case % len(shapes)]
        n = int(rng.integers(2 * 64 * w * b, 2 * 64 * w * b + 3000))
        scale = 10.0 ** rng.uniform(-2, 5)
        x0 = (rng.uniform(-1, 1, (n, 4)) * scale).astype(np.float32)
        x0[:, 3] = (10.0 ** rng.uniform(-3, 9, n)).astype(np.float32)
        x0[rng.integers(0, n, 5), 3] = 0.0                      # massless bodies
        x0[rng.integers(0, n, 3), 3] *= -1.0                    # and a few negative masses: plain arithmetic, no special cases
        x0[7, :3] = x0[3, :3]                                   # coincident pair: softening keeps it finite
        eps2 = float(np.float32((0.045 * scale / 1e5) ** 2 + 1e-12)) if case % 3 else 0.002
        ctx = nb.engine.Context(eps2=eps2, kernel=nb.KERNEL_SYMMETRIC)
        ctx.set_symmetric_shape(w, b)
        x = torch.from_numpy(x0).cuda()
        a = torch.zeros_like(x)
        ctx.accel_range(x, a, 0, n, 0, n)
        truth = oracle.accel_range(x0, 0, n, eps2=eps2, f64acc=True)
        amax = np.abs(truth[:, :3]).max()
        ctx.sync()
        assert np.abs(a.cpu().numpy() - truth)[:, :3].max() / amax <= 2e-5, (case, n, w, b, scale)
        # a random split into targets [i0,i1) and a wrapped run of everything else
        i0 = int(rng.integers(0, n - 10)); i1 = int(rng.integers(i0 + 1, min(n, i0 + n // 2) + 1))
        ai = torch.zeros((i1 - i0, 4), device="cuda")
        aj = torch.zeros((n - (i1 - i0), 4), device="cuda")
        ctx.accel_cross(x, ai, i0, i1, False, i1 % n, n - (i1 - i0), aj)
        ctx.sync()
        run = (np.arange(n - (i1 - i0)) + i1) % n
        xs = np.ascontiguousarray(np.concatenate([x0[i0:i1], x0[run]]))
        ni = i1 - i0
        want_i = oracle.accel_range(xs, 0, ni, ni, n, eps2=eps2, f64acc=True)
        want_j = oracle.accel_range(xs, ni, n, 0, ni, eps2=eps2, f64acc=True)
        gi, gj = ai.cpu().numpy(), aj.cpu().numpy()
        assert np.abs(gi - want_i)[:, :3].max() <= 2e-5 * max(np.abs(want_i[:, :3]).max(), amax * 1e-3), (case, "i")
        assert np.abs(gj - want_j)[:, :3].max() <= 2e-5 * max(np.abs(want_j[:, :3]).max(), amax * 1e-3), (case, "j")
        mi, mj = xs[:ni, 3:4].astype(np.float64), xs[ni:, 3:4].astype(np.float64)
        net = np.abs((mi * gi[:, :3]).sum(0) + (mj * gj[:, :3]).sum(0)).max()
        assert net <= 1e-5 * ((np.abs(mi) * np.abs(gi[:, :3])).sum() + (np.abs(mj) * np.abs(gj[:, :3])).sum()), case


def test_accel_cross_rejects_overlap_and_strict(nb):
    x = torch.zeros((1000, 4), device="cuda")
    a = torch.zeros((100, 4), device="cuda")
    b = torch.zeros((300, 4), device="cuda")
    ctx = nb.engine.Context()
    with pytest.raises(nb.NBodyError):
        ctx.accel_cross(x, a, 100, 200, False, 150, 300, b)      # run starts inside the targets
    with pytest.raises(nb.NBodyError):
        ctx.accel_cross(x, a, 100, 200, False, 900, 300, b)      # run wraps into the targets
    ctx.accel_cross(x, a, 100, 200, False, 200, 300, b)          # adjacent is fine
    ctx.accel_cross(x, a, 100, 200, False, 950, 150, b[:150])    # wraps to [0,100): touches nothing
    strict = nb.engine.Context(kernel=nb.KERNEL_STRICT)
    with pytest.raises(nb.NBodyError):
        strict.accel_cross(x, a, 100, 200, False, 200, 300, b)
    ctx.sync()


def test_symmetric_shape_errors(nb):
    ctx = nb.engine.Context()
    with pytest.raises(nb.NBodyError):
        ctx.set_symmetric_shape(3, 8)
    with pytest.raises(nb.NBodyError):
        ctx.set_symmetric_shape(4, 2)
    with pytest.raises(nb.NBodyError):
        ctx.set_symmetric_shape(2, 12)
    ctx.set_symmetric_shape(0, 8)
    ctx.set_symmetric_shape(0, 0)


# ---- T3 / T4: trajectories -------------------------------------------------------------------------

def test_fast_trajectory_plummer_k100(nb):
    g = load_golden("jacobi_plummer_n1024_dt0.01.npz")
    for K, tol in ((10, 2e-6), (100, 1e-5)):
        x, v, a = _gpu_run(nb, g["x0"], K, 0.01, 0.002, nb.KERNEL_FAST)
        assert np.abs(x - g[f"x_{K}"])[:, :3].max() <= tol, K
        assert nb.engine.verify_still_bodies(x, g[f"x_{K}"]) == 0


@pytest.mark.parametrize("kernel,shape", [("fast", None), ("onesided", None), ("symmetric", (1, 2)), ("symmetric", (2, 4))])
def test_benchmark_time_step_vs_the_reference_build(nb, oracle, kernel, shape):
    """The benchmark's dt = 0.01 against a REFERENCE-held fixture (tests/golden/ref_cpu_plummer_n1024_dt0.01.npz: the reference's
    CPU_compute built with DT 0.01f, validation.cpp:43-49). One step: positions within 1e-6 of the Plummer scale radius and inside
    the reference's own 1 % rule (validation.cpp:143-164); accelerations within the reference's in-place ordering effect (an
    oracle-vs-oracle quantity, computed here) plus 1e-5 of max|a|. K = 10: 1 % rule clean, 1e-5 of scale. K = 100: the in-place
    ordering of the CPU reference — not the GPU — puts 24 of 1024 bodies outside 1 % against our own Jacobi oracle (measured on
    the CPU); the GPU must stay within that effect: <= 48 offenders and 1e-2 of scale (SURVEY.md A.3 tier 4)."""
    g = load_golden("ref_cpu_plummer_n1024_dt0.01.npz")
    k = {"fast": nb.KERNEL_FAST, "onesided": nb.KERNEL_ONESIDED, "symmetric": nb.KERNEL_SYMMETRIC}[kernel]

    def run(K):
        sim = nb.engine.Simulation(g["x0"], dt=0.01, eps2=0.002, kernel=k)
        if shape:
            sim.ctx.set_symmetric_shape(*shape)
        sim.run(K)
        return sim.state()
    x, v, a = run(1)
    assert np.abs(x - g["x_1"])[:, :3].max() <= 1e-6
    assert nb.engine.verify_still_bodies(x, g["x_1"]) == 0
    aj = oracle.accel_range(g["x0"], 0, 1024, eps2=0.002)          # Jacobi order, same arithmetic
    amax = np.abs(g["a_1"][:, :3]).max()
    ordering = np.abs(aj - g["a_1"])[:, :3]
    assert np.all(np.abs(a - g["a_1"])[:, :3] <= ordering + 1e-5 * amax)
    assert np.all(ordering[0] == 0) and np.abs(a[0] - g["a_1"][0])[:3].max() <= 1e-5 * amax    # body 0 sees no ordering effect
    x10, _, _ = run(10)
    assert nb.engine.verify_still_bodies(x10, g["x_10"]) == 0 and np.abs(x10 - g["x_10"])[:, :3].max() <= 1e-5
    x100, _, _ = run(100)
    bad = nb.engine.verify_still_bodies(x100, g["x_100"])
    assert bad <= 48 and np.abs(x100 - g["x_100"])[:, :3].max() <= 1e-2, bad


@pytest.mark.parametrize("n", [65536, 262144, 1048576])
def test_baseline_configs_at_full_size_vs_the_reference_build(nb, oracle, n):
    """BASELINE configs[1], configs[2] and the system of configs[3] (N = 1048576, here on one GPU) AT FULL SIZE against the reference itself: one step (dt = 0.01) of the bench's own bodies
    through the default kernels (unit runs at 65536, block pairs at 262144, equal-mass path) and through the general path, against
    the REFERENCE build's CPU_compute on 2048 sampled bodies (tests/golden/ref_cpu_plummer_n<N>_dt0.01_sample.npz: 9 s / 2.5 min / 50 min
    of one core, generated once). The GPU may differ from the reference only by the reference's own two artefacts, both measured here
    with the pinned restatement: its in-place ORDER (bodies before i are already advanced when i is evaluated: |a_jacobi - a_ref|,
    up to 9e-5 of max|a| for the last bodies) and the ROUNDING of its N-term fp32 running sum (|a_jacobi - a_truth|, 1-2e-5) — plus
    1e-5 of max|a|. Positions and velocities follow (v = dt/2 a, x = x0 + dt v)."""
    g = load_golden(f"ref_cpu_plummer_n{n}_dt0.01_sample.npz")
    assert int(g["n"]) == n
    x0 = nb.engine.seeded_bodies(n, 1, 12345)
    assert np.array_equal(bits(x0[:8]), bits(g["x0_head"]))
    idx = g["idx"]
    for eq in (-1, 0):
        sim = nb.engine.Simulation(x0, dt=0.01, eps2=0.002)
        sim.ctx.set_equal_mass(eq)
        sim.run(1)
        x, v, a = sim.state()
        assert np.abs(x[idx] - g["x_1"])[:, :3].max() <= 1e-6
        for i0, i1 in ((0, 256), (n - 256, n)):              # the contiguous parts of the sample: first and last bodies of the in-place loop
            sel = np.searchsorted(idx, np.arange(i0, i1))
            aj = oracle.accel_range(x0, i0, i1, 0, n, eps2=0.002)                  # fp32 sequential, Jacobi order
            at = oracle.accel_range(x0, i0, i1, 0, n, eps2=0.002, f64acc=True)     # fp64-accumulated
            ar = g["a_1"][sel]
            amax = np.abs(at[:, :3]).max()
            bound = np.abs(aj - ar)[:, :3] + np.abs(aj - at)[:, :3] + 1e-5 * amax
            assert np.all(np.abs(a[i0:i1] - ar)[:, :3] <= bound), (n, eq, i0)
            assert np.all(np.abs(v[i0:i1] - g["v_1"][sel])[:, :3] <= 0.5 * 0.01 * bound + 1e-12)
            assert np.abs(a[i0:i1] - at)[:, :3].max() / amax <= 1e-5              # and the GPU itself is closer to the truth than the reference
        # the scattered part of the sample: within the largest artefact seen on the contiguous parts + the same margin
        rest = np.abs(a[idx] - g["a_1"])[:, :3].max() / np.abs(g["a_1"][:, :3]).max()
        assert rest <= (2e-4 if n <= 262144 else 5e-4), rest
        if "x_10" in g.files:       # configs[1]: a K = 10 prefix of its 1000 steps, still against the REFERENCE build. Its in-place order
            sim.run(9)              # (not the GPU) separates the two by 4.2e-7 in position and 9.7e-6 in velocity (measured on the CPU between
            x, v, a = sim.state()   # our two oracles); the reference's own 1 % rule (validation.cpp:143-164) holds for every sampled body
            assert np.abs(x[idx] - g["x_10"])[:, :3].max() <= 2e-6 and np.abs(v[idx] - g["v_10"])[:, :3].max() <= 5e-5
            assert nb.engine.verify_still_bodies(np.ascontiguousarray(x[idx]), np.ascontiguousarray(g["x_10"])) == 0


def test_reference_one_percent_rule_vs_literal_reference(nb):
    """compareHostToDevice's acceptance rule (validation.cpp:84-86, 143-164) against the literal
    reference outputs. After 10 steps the GPU passes it outright; after 100 steps the reference's
    own in-place ordering (not the GPU) moves a few % of bodies past 1 %."""
    g = load_golden("ref_cpu_n1024.npz")
    x10, _, _ = _gpu_run(nb, g["x0"], 10, 0.1, 0.002, nb.KERNEL_FAST)
    assert nb.engine.verify_still_bodies(x10, g["x_10"]) == 0
    assert np.abs(x10 - g["x_10"])[:, :3].max() / 1e5 <= 1e-4
    x100, _, _ = _gpu_run(nb, g["x0"], 100, 0.1, 0.002, nb.KERNEL_FAST)
    bad = nb.engine.verify_still_bodies(x100, g["x_100"])
    assert bad <= 64, bad   # ~3 % of 1024; caused by close encounters amplifying the ordering difference


# ---- the reference's shipped configuration: N_BODIES 8192, DT 0.1f, unseeded init ----

def _ref_n8192_start(nb):
    import ctypes
    ctypes.CDLL(None).srand(1)                       # an unseeded process starts in this state
    return nb.engine.libc_random_bodies(8192)


def test_reference_shipped_size_strict_bitwise(nb, oracle):
    """constants.h:13,25-26 through the strict kernel: bit-identical to the Jacobi oracle after 1 and 3 steps."""
    x0 = _ref_n8192_start(nb)
    assert np.array_equal(bits(x0[:8]), bits(load_golden("ref_cpu_n8192.npz")["x0_head"]))
    xo, vo, ao = x0.copy(), np.zeros_like(x0), np.zeros_like(x0)
    for K, more in ((1, 1), (3, 2)):
        oracle.step_jacobi(xo, ao, vo, dt=0.1, eps2=0.002, steps=more)
        x, v, a = _gpu_run(nb, x0, K, 0.1, 0.002, nb.KERNEL_STRICT)
        assert same_bits(x, xo) and same_bits(v, vo) and same_bits(a, ao), K


@pytest.mark.parametrize("kernel", ["fast", "onesided", "symmetric"])
def test_reference_shipped_size_fast_vs_literal_reference(nb, oracle, kernel):
    """The fast kernels at the reference's own size. Against the Jacobi oracle (same arithmetic, race-free
    order): positions to 1e-6 of the scale after 1 step, 1e-5 after 10, and the reference's 1 % rule passes
    for every body. Against the LITERAL reference outputs (golden from the reference build; its CPU loop
    advances body i before body i+1 reads it): at this size that in-place ordering alone moves two bodies by
    0.1-0.4 position units in the first step and puts 11 of 8192 bodies outside its own 1 % rule after ten
    (oracle vs oracle, computed here) — the GPU is within that ordering effect plus 1e-6 of the scale, and
    fails the rule for no more bodies than the Jacobi oracle does (+2)."""
    g = load_golden("ref_cpu_n8192.npz")
    x0 = _ref_n8192_start(nb)
    k = {"fast": nb.KERNEL_FAST, "onesided": nb.KERNEL_ONESIDED, "symmetric": nb.KERNEL_SYMMETRIC}[kernel]
    sim = nb.engine.Simulation(x0, dt=0.1, eps2=0.002, kernel=k)
    info = sim.ctx.step_info(8192)     # FAST at this size: the fused one-launch step; SYMMETRIC: block pairs; ONESIDED: LDS-tiled one-sided
    assert info["symmetric"] == (kernel == "symmetric") and info["fused"] == (kernel == "fast")
    xo, vo, ao = x0.copy(), np.zeros_like(x0), np.zeros_like(x0)
    oracle.step_jacobi(xo, ao, vo, dt=0.1, eps2=0.002, steps=1)
    sim.run(1)
    x, v, a = sim.state()
    assert np.abs(x - xo)[:, :3].max() / 1e5 <= 1e-6
    assert np.abs(a - ao)[:, :3].max() / np.abs(ao[:, :3]).max() <= 1e-5
    ordering_x = np.abs(xo - g["x_1"])[:, :3]                      # in-place vs Jacobi, per component
    assert np.all(np.abs(x - g["x_1"])[:, :3] <= ordering_x + 1e-6 * 1e5)
    assert nb.engine.verify_still_bodies(x, g["x_1"]) == 0         # the reference's own acceptance rule
    amax = np.abs(g["a_1"][:, :3]).max()
    ordering_a = np.abs(ao - g["a_1"])[:, :3]
    assert np.all(np.abs(a - g["a_1"])[:, :3] <= ordering_a + 1e-5 * amax)
    oracle.step_jacobi(xo, ao, vo, dt=0.1, eps2=0.002, steps=9)
    sim.run(9)
    x10, _, _ = sim.state()
    assert np.abs(x10 - xo)[:, :3].max() / 1e5 <= 1e-5
    assert nb.engine.verify_still_bodies(x10, xo) == 0
    assert nb.engine.verify_still_bodies(x10, g["x_10"]) <= nb.engine.verify_still_bodies(xo, g["x_10"]) + 2


# ---- the drop-in boundary ---------------------------------------------------------------------------

def test_simulate_dropin(nb, oracle):
    """simulate(d_bodies, d_acc, d_vel, N) of kernel.cuh:2: in place, synchronous, DT/EPS2 of
    constants.h, N may be smaller than the allocation."""
    n_alloc, n = 1500, 1234
    x0 = nb.engine.seeded_bodies(n_alloc, 0, 3)
    dev = torch.device("cuda", 0)
    x = torch.from_numpy(x0).to(dev)
    v = torch.zeros_like(x)
    a = torch.full_like(x, 7.0)                                     # pure output: must be overwritten
    for _ in range(3):
        nb.engine.simulate(x, a, v, n)                              # synchronous: no explicit sync
    xo, vo, ao = x0[:n].copy(), np.zeros((n, 4), np.float32), np.zeros((n, 4), np.float32)
    oracle.step_jacobi(xo, ao, vo, dt=0.1, eps2=0.002, steps=3)
    xg, vg, ag = x.cpu().numpy(), v.cpu().numpy(), a.cpu().numpy()
    assert np.abs(xg[:n] - xo)[:, :3].max() / 1e5 <= 1e-6
    assert np.abs(ag[:n] - ao)[:, :3].max() / np.abs(ao[:, :3]).max() <= 1e-5
    assert np.array_equal(xg[n:], x0[n:]) and np.all(ag[n:] == 7.0) and np.all(vg[n:] == 0)  # tail untouched
    assert torch.cuda.current_device() == dev.index                 # the caller's device is left as it was


def test_error_paths(nb):
    ctx = nb.engine.Context()
    with pytest.raises(nb.NBodyError):
        ctx.set_params(0.1, 0.0)           # eps2 must be > 0
    with pytest.raises(nb.NBodyError):
        ctx.set_kernel(nb.KERNEL_FAST, tile=100)
    with pytest.raises(nb.NBodyError):
        ctx.set_kernel(7)
    ctx.set_kernel(nb.KERNEL_ONESIDED)
    ctx.set_kernel(nb.KERNEL_SYMMETRIC)
    ctx.set_kernel(nb.KERNEL_FAST)
    x = torch.zeros((8, 4), device="cuda")
    with pytest.raises(ValueError):
        ctx.step(x, x[:4], x)
    with pytest.raises(ValueError):
        ctx.step(x.cpu(), x.cpu(), x.cpu())
    ctx.step(x[:0], x[:0], x[:0])          # empty system is a no-op
    ctx.sync()


# ---- partial ranges (what the sharded step is built from) ---------------------------------------------

@pytest.mark.parametrize("kernel", ["fast", "strict"])
def test_accel_range_blocks_compose(nb, oracle, kernel):
    n = 2500
    x0 = nb.engine.seeded_bodies(n, 0, 21)
    k = nb.KERNEL_FAST if kernel == "fast" else nb.KERNEL_STRICT
    ctx = nb.engine.Context(kernel=k)
    x = torch.from_numpy(x0).cuda()
    i0, i1 = 700, 1900
    a = torch.zeros((i1 - i0, 4), device="cuda")
    # canonical block order with accumulate: [0,i0) then [i0,i1) then [i1,n)
    ctx.accel_range(x, a, i0, i1, 0, i0, accumulate=False)
    ctx.accel_range(x, a, i0, i1, i0, i1, accumulate=True)
    ctx.accel_range(x, a, i0, i1, i1, n, accumulate=True)
    ctx.sync()
    want = oracle.accel_range(x0, i0, i1, 0, n, eps2=0.002)
    got = a.cpu().numpy()
    if kernel == "strict":
        assert same_bits(got, want)       # the sequential sum continues across calls exactly
    else:
        assert np.abs(got - want)[:, :3].max() / np.abs(want[:, :3]).max() <= 1e-5
    # a sub-range of sources only
    ctx.accel_range(x, a, i0, i1, 100, 333, accumulate=False)
    ctx.sync()
    want = oracle.accel_range(x0, i0, i1, 100, 333, eps2=0.002)
    got = a.cpu().numpy()
    assert (same_bits(got, want) if kernel == "strict" else
            np.abs(got - want)[:, :3].max() / np.abs(want[:, :3]).max() <= 1e-5)
    # empty source range writes zeros
    ctx.accel_range(x, a, i0, i1, 5, 5, accumulate=False)
    ctx.sync()
    assert np.all(a.cpu().numpy() == 0)


def test_integrate_range_matches_oracle_bitwise(nb, oracle):
    n = 1000
    rng = np.random.default_rng(8)
    x0 = _rand_bodies(n, 8)
    v0 = rng.normal(0, 50, (n, 4)).astype(np.float32); v0[:, 3] = 0
    a0 = rng.normal(0, 20, (n, 4)).astype(np.float32); a0[:, 3] = 0
    ctx = nb.engine.Context(dt=0.1)
    x = torch.from_numpy(x0).cuda()
    i0, i1 = 123, 877
    v = torch.from_numpy(v0[i0:i1].copy()).cuda()
    a = torch.from_numpy(a0[i0:i1].copy()).cuda()
    ctx.integrate_range(x, v, a, i0, i1)
    ctx.sync()
    xo, vo = x0.copy(), v0.copy()
    xs, vs = xo[i0:i1].copy(), vo[i0:i1].copy()
    oracle.integrate(xs, vs, a0[i0:i1].copy(), dt=0.1)
    xo[i0:i1] = xs
    assert same_bits(x.cpu().numpy(), xo) and same_bits(v.cpu().numpy(), vs)


# ---- full-size, size-independent properties -----------------------------------------------------------

def test_large_n_subset_against_oracle_and_properties(nb, oracle):
    """N=65536 (BASELINE configs[1]): 2048 sampled targets against ALL sources on the CPU; exact
    mass linearity; momentum balance (sum m_i a_i = 0 by Newton's third law)."""
    n = 65536
    x0 = nb.engine.seeded_bodies(n, 1, 2025)
    ctx = nb.engine.Context(dt=0.01)
    x = torch.from_numpy(x0).cuda()
    a = torch.zeros_like(x)
    ctx.accel_range(x, a, 0, n, 0, n)
    ctx.sync()
    ag = a.cpu().numpy()
    for (i0, i1) in ((0, 1024), (40000, 41024)):
        # vs the fp64-accumulated truth: 1e-5; vs the fp32 sequential oracle, whose own 65536-term
        # running sum carries ~sqrt(N)*2^-24 of rounding, 5e-5
        truth = oracle.accel_range(x0, i0, i1, 0, n, eps2=0.002, f64acc=True)
        seq32 = oracle.accel_range(x0, i0, i1, 0, n, eps2=0.002)
        amax = np.abs(truth[:, :3]).max()
        assert np.abs(ag[i0:i1] - truth)[:, :3].max() / amax <= 1e-5
        assert np.abs(ag[i0:i1] - seq32)[:, :3].max() / amax <= 5e-5
        assert np.abs(ag[i0:i1] - truth)[:, :3].max() <= np.abs(seq32 - truth)[:, :3].max()  # tiling sums better
    # momentum balance, relative to sum m|a|
    m = x0[:, 3:4].astype(np.float64)
    net = np.abs((m * ag[:, :3]).sum(0)).max()
    assert net / (m * np.abs(ag[:, :3])).sum() < 1e-6
    # doubling every mass doubles every acceleration exactly (power-of-two scaling is exact in fp32)
    x2 = x0.copy(); x2[:, 3] *= 2
    a2 = torch.zeros_like(x)
    ctx.accel_range(torch.from_numpy(x2).cuda(), a2, 0, n, 0, n)
    ctx.sync()
    assert np.array_equal(a2.cpu().numpy()[:, :3], 2 * ag[:, :3])


def test_large_n_strict_equals_fast_within_tolerance(nb):
    """One step at N=32768 with the reference's own distribution. (A second step is not compared
    on accelerations: the closest pairs reach |a| ~ 1e6 and amplify last-bit differences of step 1
    by orders of magnitude — the chaos SURVEY.md 7.2 describes — positions still agree.)"""
    n = 32768
    x0 = nb.engine.seeded_bodies(n, 0, 77)
    xs, vs, as_ = _gpu_run(nb, x0, 1, 0.1, 0.002, nb.KERNEL_STRICT)
    xf, vf, af = _gpu_run(nb, x0, 1, 0.1, 0.002, nb.KERNEL_FAST)
    norm = np.maximum(np.linalg.norm(as_[:, :3], axis=1), np.median(np.linalg.norm(as_[:, :3], axis=1)))
    assert (np.linalg.norm((af - as_)[:, :3], axis=1) / norm).max() <= 5e-5   # the strict side is a 32768-term fp32 running sum
    assert np.abs(xf - xs)[:, :3].max() / 1e5 <= 1e-6
    # second step: the bodies in the closest encounters (|a| ~ 1e6, i.e. 4.5e3 position units per
    # step) amplify last-bit differences, so compare all but the 0.1 % most accelerated bodies
    xs2, _, _ = _gpu_run(nb, x0, 2, 0.1, 0.002, nb.KERNEL_STRICT)
    xf2, _, _ = _gpu_run(nb, x0, 2, 0.1, 0.002, nb.KERNEL_FAST)
    an = np.linalg.norm(as_[:, :3], axis=1)
    calm = an <= np.quantile(an, 0.999)
    assert np.abs(xf2 - xs2)[calm][:, :3].max() / 1e5 <= 1e-6


# ---- fp64 variant ---------------------------------------------------------------------------------------

def test_f64_step_matches_oracle(nb, oracle):
    n = 2000
    x0 = nb.engine.seeded_bodies(n, 1, 4).astype(np.float64)
    xo, vo, ao = x0.copy(), np.zeros_like(x0), np.zeros_like(x0)
    oracle.step_jacobi_f64(xo, ao, vo, dt=0.01, eps2=0.002, steps=3)
    ctx = nb.engine.Context()
    x = torch.from_numpy(x0).cuda()
    v = torch.zeros_like(x)
    a = torch.zeros_like(x)
    ctx.step_f64(x, a, v, dt=0.01, eps2=0.002, steps=3)
    ctx.sync()
    assert np.abs(a.cpu().numpy() - ao)[:, :3].max() / np.abs(ao[:, :3]).max() <= 1e-12
    assert np.abs(x.cpu().numpy() - xo)[:, :3].max() <= 1e-13
    # and the fp32 fast path agrees with fp64 to fp32 accuracy (BASELINE configs[4]'s tolerance check)
    xf, _, af = _gpu_run(nb, x0.astype(np.float32), 3, 0.01, 0.002, nb.KERNEL_FAST)
    assert np.abs(xf - xo)[:, :3].max() <= 2e-6


def test_config5_f64_at_n262144(nb, oracle):
    """configs[4] at its stated size: N=262144 fp64 on one GPU. Sampled targets against the checker's all-double
    sum over all sources (1e-12 of max|a|), momentum balance in double, and the tolerance check against the fp32
    engine from the same start after 4 steps (2e-6 of the Plummer scale radius). The fp64 kernel is the build's
    own: it has no reference counterpart, so this parity is against the checker only ("parity unpinned")."""
    n = 262144
    x0 = nb.engine.seeded_bodies(n, 1, 12345)
    x64 = x0.astype(np.float64)
    ctx = nb.engine.Context()
    x = torch.from_numpy(x64).cuda()
    v = torch.zeros_like(x)
    a = torch.zeros_like(x)
    ctx.step_f64(x, a, v, dt=0.01, eps2=0.002, steps=1)
    ctx.sync()
    ag = a.cpu().numpy()
    for i0 in (0, 131000):
        want = oracle.accel_range_f64(x64, i0, i0 + 256, 0, n, eps2=0.002)
        assert np.abs(ag[i0:i0 + 256] - want)[:, :3].max() / np.abs(want[:, :3]).max() <= 1e-12
    m = x64[:, 3:4]
    assert np.abs((m * ag[:, :3]).sum(0)).max() / (m * np.abs(ag[:, :3])).sum() < 1e-13
    ctx.step_f64(x, a, v, dt=0.01, eps2=0.002, steps=3)
    ctx.sync()
    xf, _, _ = _gpu_run(nb, x0, 4, 0.01, 0.002, nb.KERNEL_FAST)
    assert np.abs(xf - x.cpu().numpy())[:, :3].max() <= 2e-6


# ---- hipGraph replay of launch-bound steps ---------------------------------------------------------------

@pytest.mark.parametrize("n,kernel", [(4096, "fast"), (1000, "fast"), (2048, "strict"), (4096, "symmetric")])
def test_graph_replay_is_identical_to_eager_launches(nb, n, kernel):
    k = {"fast": nb.KERNEL_FAST, "strict": nb.KERNEL_STRICT, "symmetric": nb.KERNEL_SYMMETRIC}[kernel]
    x0 = nb.engine.seeded_bodies(n, 1, 3)
    out = []
    for mode in (0, 1):
        sim = nb.engine.Simulation(x0, dt=0.01, eps2=0.002, kernel=k)
        sim.ctx.set_graph(mode)
        sim.run(70)              # two 32-step graph launches + 6 eager steps when the graph is on
        sim.run(40)              # cached graph reused
        out.append(sim.state())
    for p, q in zip(out[0], out[1]):
        assert np.array_equal(p, q)


def test_graph_replay_and_equal_mass_path_do_not_depend_on_call_granularity(nb):
    """A context with graph replay ON never takes the equal-mass path, whatever the number of steps of a call: 64 steps in one call
    (two replayed graphs of 32), as 40 + 24 (one graph + eager steps) and as 64 single-step calls give the same bits, with
    nbody_ctx_set_equal_mass(1) asking for the scan from 4096 bodies and every body carrying 1/N."""
    n = 10000
    x0 = nb.engine.seeded_bodies(n, 1, 5)              # Plummer: every mass 1/N
    outs = []
    for calls in ((64,), (40, 24), (1,) * 64):
        sim = nb.engine.Simulation(x0, dt=0.01, eps2=0.002)
        sim.ctx.set_graph(1)
        sim.ctx.set_equal_mass(1)
        for k in calls:
            sim.run(k)
        outs.append(sim.state())
        assert not sim.ctx.equal_mass_verdict()["scanned"]
    for other in outs[1:]:
        for p, q in zip(outs[0], other):
            assert np.array_equal(p, q)
    # graph replay off (the default): the path is taken, in one call or in many — and the bits agree as well
    eq = []
    for calls in ((64,), (1,) * 64):
        sim = nb.engine.Simulation(x0, dt=0.01, eps2=0.002)
        sim.ctx.set_equal_mass(1)
        for k in calls:
            sim.run(k)
        eq.append(sim.state())
        v = sim.ctx.equal_mass_verdict()
        assert v["scanned"] and v["uniform"]
    for p, q in zip(eq[0], eq[1]):
        assert np.array_equal(p, q)


def test_small_n_shapes_cover_the_chip(nb):
    ctx = nb.engine.Context()
    assert ctx.launch_info(8192, 8192)["blocks"] >= 512      # the reference's N_BODIES
    assert ctx.launch_info(262144, 262144)["blocks"] >= 4096
    assert ctx.launch_info(1, 1)["blocks"] == 1


# ---- the older host-pointer boundary ------------------------------------------------------

def test_legacy_host_pointer_simulate(nb, oracle):
    """simulate(float4* bodies, float3* acc, float3* vel, N) of the older snapshot: host arrays in/out,
    float3 velocity, double-literal DT/EPS2 arithmetic. Strict kernel: bit-exact vs the restatement of
    Sim-Without-OpenGL-Integration/kernel.cu:5-82 (Jacobi order); fast kernel: tolerance."""
    import ctypes as C
    n = 1500
    x0 = nb.engine.seeded_bodies(n, 1, 8)
    xo, vo = x0.copy(), np.zeros((n, 3), np.float32)
    oracle.step_legacy(xo, vo, steps=3)
    lib = nb.load()
    ctx = C.c_void_p()
    nb._lib.check(lib.nbody_default_ctx(C.byref(ctx)))
    for kernel, exact in ((nb.KERNEL_STRICT, True), (nb.KERNEL_FAST, False)):
        nb._lib.check(lib.nbody_ctx_set_kernel(ctx, kernel, 0, 0, 0))
        x, a3, v3 = x0.copy(), np.full((n, 3), 5.0, np.float32), np.zeros((n, 3), np.float32)
        for _ in range(3):
            nb.engine.simulate_host_legacy(x, a3, v3)
        if exact:
            assert same_bits(x, xo) and np.array_equal(v3.view(np.uint32), vo.view(np.uint32))
        else:
            assert np.abs(x - xo)[:, :3].max() <= 1e-6 and np.abs(v3 - vo).max() <= 1e-5 * np.abs(vo).max() + 1e-7
        assert np.all(a3 == 5.0)                     # accelerations are never copied back (kernel.cu:115-124)
        assert np.array_equal(x[:, 3], x0[:, 3])
    nb._lib.check(lib.nbody_ctx_set_kernel(ctx, nb.KERNEL_FAST, 0, 0, 0))


# ---- BASELINE.json sizes ------------------------------------------------------------------------------------

def test_config2_n65536_k10_prefix_against_oracle_subset(nb, oracle):
    """configs[1]: N=65536, dt=0.01. After 9 GPU steps take the positions; the 10th step's stored
    accelerations must be the oracle's accelerations at those positions (4096 sampled targets against
    all sources), and the 10th step's integrate must be the oracle's integrate, bit for bit."""
    n = 65536
    x0 = nb.engine.seeded_bodies(n, 1, 12345)
    sim = nb.engine.Simulation(x0, dt=0.01, eps2=0.002)
    sim.run(9)
    x9, v9, _ = sim.state()
    sim.run(1)
    x10, v10, a10 = sim.state()
    idx = [(0, 2048), (30000, 32048)]
    for i0, i1 in idx:
        truth = oracle.accel_range(x9, i0, i1, 0, n, eps2=0.002, f64acc=True)
        assert np.abs(a10[i0:i1] - truth)[:, :3].max() / np.abs(truth[:, :3]).max() <= 1e-5
    xs, vs = x9.copy(), v9.copy()
    oracle.integrate(xs, vs, a10, dt=0.01)          # same arithmetic as validation.cpp:43-49
    assert same_bits(xs, x10) and same_bits(vs, v10)
    assert np.isfinite(x10).all() and np.array_equal(x10[:, 3], x0[:, 3])


def _energy_terms(x, v, eps2):
    """Kinetic energy and softened potential energy of a state (torch fp64 on the GPU; test-side arithmetic only):
    PE = -1/2 sum_{i != j} m_i m_j / sqrt(r_ij^2 + eps2), the potential of the pair force of kernel.cu:9-29."""
    xd, vd = torch.from_numpy(x).cuda().double(), torch.from_numpy(v).cuda().double()
    m = xd[:, 3]
    ke = 0.5 * (m * (vd[:, :3] ** 2).sum(1)).sum().item()
    pe = 0.0
    for i0 in range(0, len(x), 2048):
        d = xd[i0:i0 + 2048, None, :3] - xd[None, :, :3]
        inv = torch.rsqrt((d * d).sum(2) + eps2)
        pe -= 0.5 * ((m[i0:i0 + 2048, None] * m[None, :]) * inv).sum().item()
        del d, inv
    pe += 0.5 * (m * m).sum().item() / np.sqrt(eps2)        # the i == j terms counted above
    return ke, pe


@pytest.mark.parametrize("n,path", [(65536, "runs"), (16384, "balanced"), (8192, "fused")])
def test_config2_full_run_1000_steps_conservation(nb, oracle, n, path):
    """configs[1] in full: N=65536, dt=0.01, 1000 steps from a Plummer sphere — 4.3e12 interactions, out of the CPU checker's
    reach, so the size-independent properties of the reference's own update rule (kernel.cu:116-129: v += 0.5*DT*a, x += DT*v
    — symplectic Euler for H = KE + PE/2) carry the check: KE + PE/2 stays within its O(dt) oscillation instead of drifting
    (profiles/r03_energy_probe_n*.txt: -6e-3 at the moment of collapse, step 200, within +-2e-3 afterwards, at every size from
    8192 to 262144), total momentum stays at rounding level (every pair force is applied with both signs), masses are untouched,
    and a second run from the same state is bit-identical. The same 1000 steps through the balanced runs (N=16384) and the
    fused step (N=8192, the reference's N_BODIES)."""
    dt, eps2, steps = 0.01, 0.002, 1000
    x0 = nb.engine.seeded_bodies(n, 1, 12345)
    sim = nb.engine.Simulation(x0, dt=dt, eps2=eps2)
    assert sim.ctx.step_info(n)[path]
    ke0, pe0 = _energy_terms(x0, np.zeros_like(x0), eps2)
    sim.run(steps)
    x, v, a = sim.state()
    assert np.isfinite(x).all() and np.isfinite(v).all() and np.array_equal(x[:, 3], x0[:, 3])
    ke1, pe1 = _energy_terms(x, v, eps2)
    h0, h1 = ke0 + 0.5 * pe0, ke1 + 0.5 * pe1
    assert abs(h1 - h0) <= 3e-3 * abs(h0), (ke0, pe0, ke1, pe1)
    assert ke1 > 0.05 * abs(pe0)                                  # the sphere did evolve: a cold start has fallen in
    m = x0[:, 3:4].astype(np.float64)
    p = (m * v[:, :3]).sum(0)
    assert np.abs(p).max() <= 1e-7 * (m * np.abs(v[:, :3])).sum(), p     # measured 5e-9 (profiles/r03_energy_probe_n65536.txt)
    sim2 = nb.engine.Simulation(x0, dt=dt, eps2=eps2)
    sim2.run(steps)
    for q, r in zip(sim2.state(), (x, v, a)):
        assert np.array_equal(q, r)


def test_config3_n262144_properties(nb, oracle):
    """configs[2]: N=262144 — sampled targets vs the CPU (1024 x 262144 pairs), momentum balance, exact
    mass linearity, and run-to-run bitwise reproducibility (fixed-order slab sums, no atomics)."""
    n = 262144
    x0 = nb.engine.seeded_bodies(n, 1, 12345)
    ctx = nb.engine.Context(dt=0.01)
    x = torch.from_numpy(x0).cuda()
    a = torch.zeros_like(x)
    ctx.accel_range(x, a, 0, n, 0, n)
    ctx.sync()
    ag = a.cpu().numpy()
    for i0 in (0, 200000):
        truth = oracle.accel_range(x0, i0, i0 + 512, 0, n, eps2=0.002, f64acc=True)
        assert np.abs(ag[i0:i0 + 512] - truth)[:, :3].max() / np.abs(truth[:, :3]).max() <= 1e-5
    m = x0[:, 3:4].astype(np.float64)
    assert np.abs((m * ag[:, :3]).sum(0)).max() / (m * np.abs(ag[:, :3])).sum() < 1e-6
    a2 = torch.zeros_like(x)
    ctx.accel_range(x, a2, 0, n, 0, n)
    ctx.sync()
    assert torch.equal(a, a2)
    xs = x0.copy(); xs[:, 3] *= 4
    ctx.accel_range(torch.from_numpy(xs).cuda(), a2, 0, n, 0, n)
    ctx.sync()
    assert np.array_equal(a2.cpu().numpy()[:, :3], 4 * ag[:, :3])


@pytest.mark.parametrize("kernel", ["fast", "strict"])
def test_accel_wrapped_source_run(nb, oracle, kernel):
    """Sources j0 .. j0+count-1 modulo N in one launch (the remote pass of a rank): equals the sum
    over the two plain ranges it wraps across; the strict kernel continues the running sum in that
    exact order."""
    n = 3000
    x0 = nb.engine.seeded_bodies(n, 0, 41)
    k = nb.KERNEL_FAST if kernel == "fast" else nb.KERNEL_STRICT
    ctx = nb.engine.Context(kernel=k)
    x = torch.from_numpy(x0).cuda()
    for (i0, i1) in ((1000, 1750), (0, 750), (2250, 3000)):
        a = torch.zeros((i1 - i0, 4), device="cuda")
        ctx.accel_range(x, a, i0, i1, i0, i1, False)                       # own block first
        ctx.accel_wrapped(x, a, i0, i1, i1 % n, n - (i1 - i0), True)       # then everybody else, wrapping
        ctx.sync()
        got = a.cpu().numpy()
        if kernel == "strict":
            want = oracle.accel_range(x0, i0, i1, i0, i1, eps2=0.002)
            # continue the sequential sums in the wrapped order
            for t, i in enumerate(range(i0, i1)):
                acc = want[t].copy()
                for jj in range(i1, i1 + n - (i1 - i0)):
                    acc = oracle.pair(x0[i], x0[jj % n], acc, eps2=0.002)
                want[t] = acc
            assert same_bits(got, want)
        else:
            want = oracle.accel_range(x0, i0, i1, 0, n, eps2=0.002, f64acc=True)
            assert np.abs(got - want)[:, :3].max() / np.abs(want[:, :3]).max() <= 1e-5
    with pytest.raises(nb.NBodyError):
        ctx.accel_wrapped(x, a, 0, 750, n, 10)            # j0 must lie inside the array
    with pytest.raises(nb.NBodyError):
        ctx.accel_wrapped(x, a, 0, 750, 5, n + 1)         # at most one lap


def test_randomised_ranges_strict_bitwise(nb, oracle):
    """40 random (N, target range, source range, accumulate chain) cases: the strict kernel through
    nbody_accel_range / nbody_accel_wrapped equals the oracle's sequential sums bit for bit."""
    rng = np.random.default_rng(99)
    ctx = nb.engine.Context(kernel=nb.KERNEL_STRICT)
    for case in range(40):
        n = int(rng.integers(2, 2500))
        x0 = _rand_bodies(n, 1000 + case, scale=10.0 ** rng.uniform(-2, 5), mlo=1e-3, mhi=1e9)
        x = torch.from_numpy(x0).cuda()
        i0 = int(rng.integers(0, n)); i1 = int(rng.integers(i0 + 1, n + 1))
        cuts = sorted(int(c) for c in rng.integers(0, n + 1, 3))
        a = torch.zeros((i1 - i0, 4), device="cuda")
        want = np.zeros((i1 - i0, 4), np.float32)
        first = True
        for j0, j1 in ((0, cuts[0]), (cuts[0], cuts[1]), (cuts[1], cuts[2]), (cuts[2], n)):
            ctx.accel_range(x, a, i0, i1, j0, j1, accumulate=not first)
            first = False
        ctx.sync()
        want = oracle.accel_range(x0, i0, i1, 0, n, eps2=0.002)     # the same sources in the same order
        assert same_bits(a.cpu().numpy(), want), (case, n, i0, i1, cuts)


def test_strict_kernel_on_hostile_inputs(nb, oracle):
    """Extreme magnitudes (overflowing d*d*d, denormal products), coincident bodies, zero and negative
    masses, moving bodies: strict kernel == Jacobi oracle bit for bit, NaN for NaN."""
    rng = np.random.default_rng(31415)
    for case in range(25):
        n = int(rng.integers(1, 700))
        mag = 10.0 ** rng.uniform(-18, 18)
        x0 = (rng.normal(0, 1, (n, 4)) * mag).astype(np.float32)
        x0[:, 3] = (rng.normal(0, 1, n) * 10.0 ** rng.uniform(-30, 12)).astype(np.float32)
        if n > 3:
            x0[1, :3] = x0[0, :3]
            x0[2, 3] = 0.0
        xo, vo, ao = x0.copy(), np.zeros_like(x0), np.zeros_like(x0)
        with np.errstate(all="ignore"):
            oracle.step_jacobi(xo, ao, vo, dt=0.1, eps2=0.002, steps=2)
        x, v, a = _gpu_run(nb, x0, 2, 0.1, 0.002, nb.KERNEL_STRICT)
        for p, q, nm in ((x, xo, "x"), (v, vo, "v"), (a, ao, "a")):
            same = (bits(p) == bits(q)) | ((p == 0) & (q == 0)) | (np.isnan(p) & np.isnan(q))
            assert same.all(), (case, n, mag, nm, int((~same).sum()))


def test_config4_size_n1048576_on_one_gpu(nb, oracle):
    """configs[3]'s size (N=1048576) as one rank-less launch: sampled targets against the CPU over all
    1M sources, momentum balance, and the wrapped-run form a rank of the 8-GPU run would issue."""
    n = 1048576
    x0 = nb.engine.seeded_bodies(n, 1, 4242)
    ctx = nb.engine.Context(dt=0.01)
    x = torch.from_numpy(x0).cuda()
    a = torch.zeros_like(x)
    ctx.accel_range(x, a, 0, n, 0, n)
    ctx.sync()
    ag = a.cpu().numpy()
    assert np.isfinite(ag).all()
    for i0 in (0, 777000):
        truth = oracle.accel_range(x0, i0, i0 + 256, 0, n, eps2=0.002, f64acc=True)
        assert np.abs(ag[i0:i0 + 256] - truth)[:, :3].max() / np.abs(truth[:, :3]).max() <= 2e-5
    m = x0[:, 3:4].astype(np.float64)
    assert np.abs((m * ag[:, :3]).sum(0)).max() / (m * np.abs(ag[:, :3])).sum() < 1e-6
    # rank 3 of 8: own block, then everybody else through one wrapped launch
    S = n // 8
    i0, i1 = 3 * S, 4 * S
    ar = torch.zeros((S, 4), device="cuda")
    ctx.accel_range(x, ar, i0, i1, i0, i1, False)
    ctx.accel_wrapped(x, ar, i0, i1, i1, n - S, True)
    ctx.sync()
    arn = ar.cpu().numpy()
    # two different fp32 summation orders over 1M terms each: both within 2e-5 of the truth, 4e-5 of each other
    assert np.abs(arn - ag[i0:i1])[:, :3].max() / np.abs(ag[:, :3]).max() <= 4e-5
    truth = oracle.accel_range(x0, i0, i0 + 256, 0, n, eps2=0.002, f64acc=True)
    assert np.abs(arn[:256] - truth)[:, :3].max() / np.abs(truth[:, :3]).max() <= 2e-5


def test_partial_sum_workspace_beyond_2p31_elements(nb, oracle):
    """Maximum sizes: N = 2 400 000 (ragged: not a multiple of the 2560-body block) needs 938 slabs of N float4 = 2.25e9
    partial-sum elements (36 GB) — past what a 32-bit element index reaches. One whole step: sampled bodies at both ends of
    the array against the CPU over all sources, momentum balance, every output finite, masses untouched."""
    n = 2400000
    x0 = nb.engine.seeded_bodies(n, 1, 99)
    sim = nb.engine.Simulation(x0, dt=0.01, eps2=0.002)
    info = sim.ctx.step_info(n)
    assert info["symmetric"] and not info["runs"] and info["slabs"] * n > 2 ** 31, info
    sim.run(1)
    x, v, a = sim.state()
    assert np.isfinite(a).all() and np.isfinite(x).all() and np.array_equal(x[:, 3], x0[:, 3])
    for i0 in (0, n - 128):
        truth = oracle.accel_range(x0, i0, i0 + 128, 0, n, eps2=0.002, f64acc=True)
        assert np.abs(a[i0:i0 + 128] - truth)[:, :3].max() / np.abs(truth[:, :3]).max() <= 3e-5
        vt = 0.5 * 0.01 * truth[:, :3]
        assert np.abs(v[i0:i0 + 128, :3] - vt).max() / np.abs(vt).max() <= 3e-5
    m = x0[:, 3:4].astype(np.float64)
    assert np.abs((m * a[:, :3]).sum(0)).max() / (m * np.abs(a[:, :3])).sum() < 1e-6


# ---- workspace cap and fallback (the symmetric kernels' O(N^2/B) partial-sum slabs) ---------------------------------

def test_workspace_limit_steers_the_shape_choice_and_keeps_the_result(nb, oracle):
    """nbody_ctx_set_workspace_limit: a slab decomposition whose workspace exceeds the cap is not chosen. A whole step then keeps the
    symmetric arithmetic with the sums added in place (no workspace); with that switched off (nbody_ctx_set_inplace_sums(0)) it falls
    back to a smaller footprint, finally to the one-sided kernel. The answer stays within the fast tolerances either way."""
    n = 20000
    x0 = nb.engine.seeded_bodies(n, 1, 21)
    truth = oracle.accel_range(x0, 0, 2048, 0, n, eps2=0.002, f64acc=True)
    sim = nb.engine.Simulation(x0, dt=0.01, eps2=0.002)
    free_choice = sim.ctx.step_info(n)
    assert free_choice["symmetric"] and not free_choice["ticket"]
    need = free_choice["slabs"] * n * 16
    sim.run(1)
    a_free = sim.state()[2]
    for inplace in (-1, 0):
        for limit, unconstrained_kind in ((need // 2, None), (7 * n * 16, False), (1, False)):   # the largest block (2560 bodies) needs 8 slabs
            s2 = nb.engine.Simulation(x0, dt=0.01, eps2=0.002)
            s2.ctx.set_inplace_sums(inplace)
            s2.ctx.set_workspace_limit(limit)
            info = s2.ctx.step_info(n)
            assert info["slabs"] * n * 16 <= limit or not info["symmetric"]     # the one-sided kernel's <= 64 slabs are always allowed
            if unconstrained_kind is not None:
                if inplace == 0:
                    assert not info["symmetric"] and not info["ticket"]          # the older fallback: one-sided
                else:
                    assert info["symmetric"] and info["ticket"] and info["slabs"] in (0, 2, 4, 8)   # block pairs, sums in place (lanes under the cap)
            s2.run(1)
            a = s2.state()[2]
            assert np.abs(a[:2048] - truth)[:, :3].max() / np.abs(truth[:, :3]).max() <= 1e-5
            assert np.abs(a - a_free)[:, :3].max() / np.abs(a_free[:, :3]).max() <= 1e-5
            s2.ctx.set_workspace_limit(0)                                        # automatic again: the free choice is back
            assert s2.ctx.step_info(n) == free_choice


@pytest.mark.parametrize("n,kernel", [(1300, "symmetric"), (5000, "symmetric"), (20001, "symmetric"), (70000, "fast")])
def test_in_place_block_sums_vs_oracle(nb, oracle, n, kernel):
    """nbk::force_sym_ticket — block pairs with the partial sums added straight into the acceleration array, the order fixed per block
    by a ticket: against the Jacobi oracle and fp64-accumulated truth (T2 bars), bit-identical run to run, and within rounding of the
    slab kernel (the same block sums in another fixed order). Ragged sizes: the last block is partly padding."""
    k = nb.KERNEL_SYMMETRIC if kernel == "symmetric" else nb.KERNEL_FAST
    x0 = nb.engine.seeded_bodies(n, 0 if n < 10000 else 1, 31 + n)
    dt = 0.1 if n < 10000 else 0.01
    steps = 3 if n < 30000 else 1
    runs = []
    for rep in range(2):
        sim = nb.engine.Simulation(x0, dt=dt, eps2=0.002, kernel=k)
        sim.ctx.set_inplace_sums(1)
        info = sim.ctx.step_info(n)
        assert info["ticket"] and info["symmetric"] and info["slabs"] in (0, 2, 4, 8) and info["block_bodies"] in (640, 2560), info
        sim.run(steps)
        runs.append(sim.state())
    for p, q in zip(*runs):
        assert np.array_equal(p, q)                                   # the tickets fix the order: the same bits every run
    x, v, a = runs[0]
    slab = nb.engine.Simulation(x0, dt=dt, eps2=0.002, kernel=nb.KERNEL_SYMMETRIC)
    slab.ctx.set_inplace_sums(0)
    slab.ctx.set_symmetric_shape(1 if info["block_bodies"] == 640 else 4, 10)
    assert not slab.ctx.step_info(n)["ticket"]
    slab.run(steps)
    xs, vs, as_ = slab.state()
    scale = np.abs(as_[:, :3]).max()
    assert np.abs(a - as_)[:, :3].max() / scale <= 2e-6 and np.abs(x - xs)[:, :3].max() <= 1e-6 * max(1.0, np.abs(xs[:, :3]).max())
    assert np.all(a[:, 3] == 0)
    if n <= 20001:
        xo, vo, ao = x0.copy(), np.zeros_like(x0), np.zeros_like(x0)
        oracle.step_jacobi(xo, ao, vo, dt=dt, eps2=0.002, steps=steps)
        assert np.abs(a - ao)[:, :3].max() / np.abs(ao[:, :3]).max() <= 1e-5
        assert np.abs(x - xo)[:, :3].max() <= 1e-6 * max(1.0, np.abs(xo[:, :3]).max())
    else:
        m = 1024
        truth = oracle.accel_range(x0, n - m, n, 0, n, eps2=0.002, f64acc=True)      # the ragged last block included
        sim1 = nb.engine.Simulation(x0, dt=dt, eps2=0.002, kernel=k)
        sim1.ctx.set_inplace_sums(1)
        sim1.run(1)
        a1 = sim1.state()[2]
        assert np.abs(a1[n - m:] - truth)[:, :3].max() / np.abs(truth[:, :3]).max() <= 1e-5


def test_accel_range_on_a_square_block_keeps_the_symmetric_arithmetic_under_a_cap(nb, oracle):
    """nbody_accel_range(targets == sources) — the piece the sharded step's own-block pass is made of — under a workspace cap that no slab
    decomposition fits: the block sums are added in place into a few lanes and summed (2 … 8 lanes of 16·n bytes), `accumulate` included;
    only a cap below two lanes ends in the one-sided kernel. Same tolerances as the unconstrained evaluation, bit-identical run to run."""
    n, i0 = 30000, 1000                       # a block that does not start at body 0
    nt = 24000
    x0 = nb.engine.seeded_bodies(n, 1, 77)
    x = torch.from_numpy(x0).cuda()
    truth = oracle.accel_range(x0, i0, i0 + 1024, i0, i0 + nt, eps2=0.002, f64acc=True)
    free = nb.engine.Context(dt=0.01, eps2=0.002)
    ref = torch.zeros((nt, 4), device="cuda")
    free.accel_range(x, ref, i0, i0 + nt, i0, i0 + nt)
    free.sync()
    ref = ref.cpu().numpy()
    scale = np.abs(ref[:, :3]).max()
    for limit, want_lanes in ((8 * nt * 16, 8), (5 * nt * 16, 4), (2 * nt * 16, 2)):
        ctx = nb.engine.Context(dt=0.01, eps2=0.002)
        ctx.set_workspace_limit(limit)
        js = ctx.launch_info(nt, nt)["jsplit"]
        assert js == want_lanes, (limit, js)
        outs = []
        for rep in range(2):
            out = torch.full((nt, 4), 5.0, device="cuda")
            ctx.accel_range(x, out, i0, i0 + nt, i0, i0 + nt)
            ctx.sync()
            outs.append(out.cpu().numpy())
        assert np.array_equal(outs[0], outs[1])
        assert np.abs(outs[0] - ref)[:, :3].max() / scale <= 2e-6 and np.all(outs[0][:, 3] == 0)
        assert np.abs(outs[0][:1024] - truth)[:, :3].max() / np.abs(truth[:, :3]).max() <= 1e-5
        acc = torch.from_numpy(ref.copy()).cuda()
        ctx.accel_range(x, acc, i0, i0 + nt, i0, i0 + nt, accumulate=True)          # adds to what the array holds
        ctx.sync()
        assert np.abs(acc.cpu().numpy() - 2.0 * ref)[:, :3].max() / scale <= 4e-6
    tiny = nb.engine.Context(dt=0.01, eps2=0.002)
    tiny.set_workspace_limit(nt * 16)           # not even two lanes: the one-sided kernel (its <= 64 slabs are always allowed)
    out = torch.zeros((nt, 4), device="cuda")
    tiny.accel_range(x, out, i0, i0 + nt, i0, i0 + nt)
    tiny.sync()
    assert np.abs(out.cpu().numpy() - ref)[:, :3].max() / scale <= 1e-5


def test_in_place_block_sums_abort_instead_of_hanging(nb):
    """The failure path of the ticket protocol, exercised by the library's test hook (nbody_ctx_set_inplace_sums(ctx, 2): the next
    in-place launch finds one ticket held by nobody and may wait 2 ms): the first waiter gives up, raises the abort word — every later
    wait of the launch falls through at once, so the launch ENDS — and the host-mapped error word; the next synchronisation returns an
    error that says what happened; the tickets are reset and the same context steps correctly afterwards."""
    import time
    n = 20001
    x0 = nb.engine.seeded_bodies(n, 1, 3)
    good = nb.engine.Simulation(x0, dt=0.01, eps2=0.002, kernel=nb.KERNEL_SYMMETRIC)
    good.ctx.set_inplace_sums(1)
    good.run(2)
    want = good.state()
    ctx = nb.engine.Context(dt=0.01, eps2=0.002, kernel=nb.KERNEL_SYMMETRIC)
    ctx.set_inplace_sums(2)
    assert ctx.step_info(n)["ticket"]
    x = torch.from_numpy(x0).cuda()
    v, a = torch.zeros_like(x), torch.zeros_like(x)
    t0 = time.perf_counter()
    ctx.step(x, a, v, 1)
    with pytest.raises(nb.NBodyError) as e:
        ctx.sync()
    took = time.perf_counter() - t0
    assert e.value.code == nb._lib.ERR_HIP and "waited more than" in str(e.value), str(e.value)
    assert took < 5.0, took                                    # one 2-ms time-out, then everything falls through: not one time-out per waiter
    ctx.sync()                                                 # reported once; the context is usable again
    x.copy_(torch.from_numpy(x0).cuda())                       # (that step's sums were incomplete: restore the state, step again)
    v.zero_()
    a.zero_()
    ctx.step(x, a, v, 2)
    ctx.sync()
    for got, w in zip((x, v, a), want):
        assert np.array_equal(got.cpu().numpy(), w)


def test_workspace_limit_at_a_quarter_keeps_the_symmetric_step_at_n262144(nb, oracle):
    """configs[2]'s size with the workspace capped at a quarter of the slab kernel's footprint: the step keeps the
    symmetric arithmetic — block pairs, sums in place, no workspace — instead of falling to the one-sided kernel (-30 %): within 1e-5 of
    max|a| of the CPU's fp64-accumulated sums on sampled bodies, bit-identical run to run, and its step time within a few per cent of the
    unconstrained one (measured here over 10 queued steps each, same process; the bar is loose enough for a shared box)."""
    import time
    n = 262144
    x0 = nb.engine.seeded_bodies(n, 1, 12345)
    free = nb.engine.Simulation(x0, dt=0.01, eps2=0.002)
    free.ctx.set_equal_mass(0)
    fi = free.ctx.step_info(n)
    assert fi["symmetric"] and fi["slabs"] == 103 and not fi["ticket"]
    capped = nb.engine.Simulation(x0, dt=0.01, eps2=0.002)
    capped.ctx.set_equal_mass(0)
    capped.ctx.set_workspace_limit(fi["slabs"] * n * 16 // 4)
    ci = capped.ctx.step_info(n)
    assert ci["ticket"] and ci["symmetric"] and ci["slabs"] == 8 and ci["block_bodies"] == 2560, ci      # 8 lanes = 32 MiB of the 103-MiB cap
    capped.run(1)
    a1 = capped.state()[2]
    again = nb.engine.Simulation(x0, dt=0.01, eps2=0.002)
    again.ctx.set_equal_mass(0)
    again.ctx.set_workspace_limit(fi["slabs"] * n * 16 // 4)
    again.run(1)
    assert np.array_equal(a1, again.state()[2])                       # deterministic
    m = 512
    for i0 in (0, 131072 - 256, n - m):
        truth = oracle.accel_range(x0, i0, i0 + m, 0, n, eps2=0.002, f64acc=True)
        assert np.abs(a1[i0:i0 + m] - truth)[:, :3].max() / np.abs(truth[:, :3]).max() <= 1e-5
    free.run(1)
    assert np.abs(a1 - free.state()[2])[:, :3].max() / np.abs(a1[:, :3]).max() <= 2e-6
    t = {}
    for name, s in (("free", free), ("capped", capped), ("free2", free), ("capped2", capped)):
        s.run(3, sync=False)
        s.ctx.sync()
        t0 = time.perf_counter()
        s.run(10, sync=False)
        s.ctx.sync()
        t[name] = (time.perf_counter() - t0) / 10
    ratio = min(t["capped"], t["capped2"]) / min(t["free"], t["free2"])
    print(f"N=262144 step: unconstrained {min(t['free'], t['free2']) * 1e3:.3f} ms, capped at a quarter (sums in place) {min(t['capped'], t['capped2']) * 1e3:.3f} ms, ratio {ratio:.4f}")
    assert ratio < 1.10, t


def test_failed_workspace_allocation_falls_back_instead_of_erroring(nb, oracle):
    """The allocation itself fails (test hook: every workspace above the limit 'runs out of memory'): nbody_step,
    nbody_accel_range, nbody_step_f64 and nbody_ctx_reserve lower the cap and re-resolve instead of returning an error."""
    n = 20000
    x0 = nb.engine.seeded_bodies(n, 1, 22)
    truth = oracle.accel_range(x0, 0, 1024, 0, n, eps2=0.002, f64acc=True)
    ctx = nb.engine.Context(dt=0.01, eps2=0.002)
    ctx.set_symmetric_shape(1, 2)                                   # 128-body blocks: 157 slabs
    ctx.set_workspace_limit(n * 16 * 40, fail_above=True)          # room for 40 slabs (the one-sided kernel wants 32); nothing is refused up front
    assert ctx.step_info(n)["symmetric"] and ctx.step_info(n)["slabs"] == 157   # the planner still asks for the symmetric shape ...
    x = torch.from_numpy(x0).cuda()
    v, a = torch.zeros_like(x), torch.zeros_like(x)
    ctx.step(x, a, v, 1)                                            # ... the allocation fails, the step falls back and succeeds
    ctx.sync()
    got = a.cpu().numpy()
    assert np.abs(got[:1024] - truth)[:, :3].max() / np.abs(truth[:, :3]).max() <= 1e-5
    assert ctx.step_info(n)["ticket"]                               # the cap now sits below every slab footprint: a whole step adds in place
    ctx.set_inplace_sums(0)
    assert not ctx.step_info(n)["symmetric"]                        # ... or, with that switched off, runs the one-sided kernel
    ctx.step(x, a, v, 1)
    ctx.sync()
    out = torch.zeros_like(x)
    ctx.accel_range(x, out, 0, n, 0, n)                             # the square block through nbody_accel_range: same
    ctx.sync()
    ctx.reserve(n)                                                  # and reserve() succeeds (one-sided footprint)
    # fp64: the symmetric double kernel's slabs fail the same way, the one-sided double kernel takes over
    ctx64 = nb.engine.Context(dt=0.01, eps2=0.002)
    assert ctx64.step_info_f64(n)["symmetric"]
    ctx64.set_workspace_limit(n * 32 * 70, fail_above=True)        # the one-sided f64 kernel's <= 64 slabs fit, 1 slab per block does not
    xd = torch.from_numpy(x0.astype(np.float64)).cuda()
    vd, ad = torch.zeros_like(xd), torch.zeros_like(xd)
    ctx64.step_f64(xd, ad, vd, 0.01, 0.002, 1)
    ctx64.sync()
    t64 = oracle.accel_range_f64(x0.astype(np.float64), 0, 512, 0, n, eps2=0.002)
    assert np.abs(ad.cpu().numpy()[:512] - t64)[:, :3].max() / np.abs(t64[:, :3]).max() <= 1e-12
    assert np.abs(ad.cpu().numpy()[:1024] - truth)[:, :3].max() / np.abs(truth[:, :3]).max() <= 1e-5
    assert not ctx64.step_info_f64(n)["symmetric"]


def test_accel_cross_in_pieces_under_a_workspace_limit(nb, oracle):
    """nbody_accel_cross has no one-sided fallback: under a cap (or a failed allocation) it cuts the source run into pieces and
    evaluates them one after the other. Same pairs, both sides' sums, within the fast tolerances of the uncut evaluation."""
    n, i0, i1, j0, count = 9000, 1000, 3000, 7000, 2900            # the run wraps past the end (7000..8999, 0..899)
    x0 = nb.engine.seeded_bodies(n, 0, 31)
    x = torch.from_numpy(x0).cuda()
    ref_i, ref_j = torch.zeros((i1 - i0, 4), device="cuda"), torch.zeros((count, 4), device="cuda")
    ctx = nb.engine.Context()
    ctx.accel_cross(x, ref_i, i0, i1, False, j0, count, ref_j)
    ctx.sync()
    for limit, hook in ((100_000, False), (100_000, True), (40_000, False)):   # uncut: >= 110 KB; 2 pieces; 2+ after a failed allocation; 8
        c2 = nb.engine.Context()
        c2.set_workspace_limit(limit, fail_above=hook)
        ai, aj = torch.full((i1 - i0, 4), 3.0, device="cuda"), torch.full((count, 4), 4.0, device="cuda")
        c2.accel_cross(x, ai, i0, i1, False, j0, count, aj)
        c2.sync()
        for got, want in ((ai, ref_i), (aj, ref_j)):
            g, w = got.cpu().numpy(), want.cpu().numpy()
            assert np.abs(g - w)[:, :3].max() / np.abs(w[:, :3]).max() <= 1e-5
            assert np.all(g[:, 3] == 0)
    # an absurd cap is an error with a message, not a crash — "out of workspace", reported before any sum has been touched
    c3 = nb.engine.Context()
    c3.set_workspace_limit(64)
    keep = torch.full((i1 - i0, 4), 7.0, device="cuda")
    with pytest.raises(nb.NBodyError) as e:
        c3.accel_cross(x, keep, i0, i1, True, j0, count, ref_j)
    assert e.value.code == nb._lib.ERR_NOMEM and bool((keep == 7.0).all())
    # a shape request that no built kernel can honour for these targets is a CONFIGURATION error at once, not a halving loop that
    # ends in "out of memory": 300000 targets in blocks of 128 bodies would need 2344 target blocks (limit 2048)
    big = torch.zeros((300064, 4), device="cuda")
    c4 = nb.engine.Context()
    c4.set_symmetric_shape(1, 2)
    out_i, out_j = torch.zeros((300000, 4), device="cuda"), torch.zeros((64, 4), device="cuda")
    with pytest.raises(nb.NBodyError) as e:
        c4.accel_cross(big, out_i, 0, 300000, False, 300000, 64, out_j)
    assert e.value.code == nb._lib.ERR_CONFIG


def test_fp64_only_shape_request_is_auto_for_fp32(nb, oracle):
    """(4,6) is a double-precision block shape. Set on a context that then runs fp32 work it must not silently disable the
    symmetric kernel (nor break nbody_accel_cross, which the sharded step depends on): it counts as 'auto' there."""
    n = 20000
    ctx = nb.engine.Context(dt=0.01, eps2=0.002)
    auto = ctx.step_info(n)
    ctx.set_symmetric_shape(4, 6)
    assert ctx.step_info(n) == auto and auto["symmetric"]
    assert ctx.step_info_f64(262144)["block_bodies"] == 64 * 4 * 6
    x0 = nb.engine.seeded_bodies(6000, 0, 8)
    x = torch.from_numpy(x0).cuda()
    ai, aj = torch.zeros((2000, 4), device="cuda"), torch.zeros((3000, 4), device="cuda")
    ctx.accel_cross(x, ai, 0, 2000, False, 2500, 3000, aj)
    ctx.sync()
    want = oracle.accel_range(x0, 0, 2000, 2500, 5500, eps2=0.002, f64acc=True)
    assert np.abs(ai.cpu().numpy() - want)[:, :3].max() / np.abs(want[:, :3]).max() <= 1e-5


# ---- balanced runs: the symmetric kernel cut at rotation-step granularity (small and mid N) -------------------------------

@pytest.mark.parametrize("n,bpl,init", [(128, 2, 0), (777, 2, 1), (1000, 4, 0), (3001, 4, 1), (4099, 8, 0), (6144, 10, 1), (8192, 0, 0),
                                        (9000, 8, 1), (12345, 0, 1), (20000, 10, 0)])
def test_balanced_runs_vs_oracle(nb, oracle, n, bpl, init):
    """nbody_ctx_set_symmetric_runs(2): every worker the same number of rotation steps, units split between workers anywhere,
    per-chunk inboxes, one reduce-and-integrate kernel. Ragged sizes (padding bodies), every bodies-per-lane shape, sizes where a
    worker has fewer steps than a unit (units shared by 3+ workers): accelerations against the fp64-accumulated CPU sums,
    momentum balance, w = 0, bitwise run-to-run reproducibility, and positions after the integrate against the Jacobi oracle."""
    x0 = nb.engine.seeded_bodies(n, init, 11)
    dt = 0.1 if init == 0 else 0.01
    sim = nb.engine.Simulation(x0, dt=dt, eps2=0.002, kernel=nb.KERNEL_SYMMETRIC)
    sim.ctx.set_symmetric_shape(0, bpl)
    sim.ctx.set_symmetric_runs(2)
    sim.ctx.reserve(n)
    info = sim.ctx.step_info(n)
    assert info["balanced"] and info["symmetric"] and not info["runs"]
    assert info["block_bodies"] == 64 * (bpl or info["block_bodies"] // 64)
    bi = info["block_bodies"]                       # each unordered pair once, plus the diagonal blocks both ways and the padding
    assert 0.5 * n * n <= info["evaluated_pairs"] <= 0.5 * (n + bi) ** 2 + (n + bi) * bi
    sim.run(1)
    x, v, a = sim.state()
    truth = oracle.accel_range(x0, 0, n, eps2=0.002, f64acc=True)
    amax = np.abs(truth[:, :3]).max()
    assert np.abs(a - truth)[:, :3].max() / amax <= 1e-5
    assert np.all(a[:, 3] == 0) and np.all(v[:, 3] == 0) and np.array_equal(x[:, 3], x0[:, 3])
    m = x0[:, 3:4].astype(np.float64)
    assert np.abs((m * a[:, :3]).sum(0)).max() / (m * np.abs(a[:, :3])).sum() < 1e-6
    xo, vo, ao = x0.copy(), np.zeros_like(x0), np.zeros_like(x0)
    oracle.step_jacobi(xo, ao, vo, dt=dt, eps2=0.002)
    assert np.abs(x - xo)[:, :3].max() / (1e5 if init == 0 else 1.0) <= 1e-6
    again = nb.engine.Simulation(x0, dt=dt, eps2=0.002, kernel=nb.KERNEL_SYMMETRIC)
    again.ctx.set_symmetric_shape(0, bpl)
    again.ctx.set_symmetric_runs(2)
    again.run(1)
    x2, v2, a2 = again.state()
    assert np.array_equal(a, a2) and np.array_equal(x, x2) and np.array_equal(v, v2)


def test_balanced_runs_are_what_fast_picks_at_mid_sizes(nb, oracle):
    """FAST from 12288 to 32768 bodies is the balanced-run variant (up to 10240 the fused step, 65536 unit runs); without the fused step
    it is what runs at the reference's shipped N_BODIES = 8192 (constants.h:13). Ten steps at N = 8192 from the reference's kind of initial conditions against the Jacobi
    oracle, and the square block of nbody_accel_range (accumulate on and off) through the same kernels."""
    ctx = nb.engine.Context()
    for n in (1024, 4096, 7168, 8192):
        assert ctx.step_info(n)["fused"] and not ctx.step_info(n)["symmetric"], n      # whole steps of small systems: one launch
    for n in (9216, 12288, 16384, 32768):
        assert ctx.step_info(n)["balanced"], n
    assert ctx.step_info(65536)["runs"] and ctx.step_info(262144)["symmetric"] and not ctx.step_info(262144)["balanced"]
    n = 8192
    x0 = nb.engine.seeded_bodies(n, 0, 3)
    sim = nb.engine.Simulation(x0, dt=0.1, eps2=0.002)
    sim.ctx.set_fused(0)                                          # without the fused step FAST is the balanced-run variant here
    assert sim.ctx.step_info(n)["balanced"]
    sim.run(10)
    x, v, a = sim.state()
    xo, vo, ao = x0.copy(), np.zeros_like(x0), np.zeros_like(x0)
    oracle.step_jacobi(xo, ao, vo, dt=0.1, eps2=0.002, steps=10)
    assert np.abs(x - xo)[:, :3].max() / 1e5 <= 1e-5             # ten steps: the bar of test_reference_shipped_size_*
    assert nb.engine.verify_still_bodies(x, xo) == 0
    xd = torch.from_numpy(x0).cuda()
    out = torch.full((n, 4), 7.0, device="cuda")
    ctx2 = nb.engine.Context(eps2=0.002)
    ctx2.accel_range(xd, out, 0, n, 0, n)
    ctx2.sync()
    a1 = out.cpu().numpy()
    truth = oracle.accel_range(x0, 0, n, eps2=0.002, f64acc=True)
    assert np.abs(a1 - truth)[:, :3].max() / np.abs(truth[:, :3]).max() <= 1e-5
    ctx2.accel_range(xd, out, 0, n, 0, n, accumulate=True)
    ctx2.sync()
    assert np.abs(out.cpu().numpy() - 2 * a1)[:, :3].max() / np.abs(a1[:, :3]).max() <= 1e-6


def test_balanced_runs_workspace_is_recleared_after_other_users(nb, oracle):
    """The inboxes rely on never-written records reading as zero. Anything else that uses the context's workspace in between
    (another size, the one-sided kernel on a rectangle, block pairs) must trigger a fresh clear: same bits afterwards."""
    n = 10000
    x0 = nb.engine.seeded_bodies(n, 1, 5)
    x = torch.from_numpy(x0).cuda()
    ctx = nb.engine.Context(dt=0.01, eps2=0.002)
    assert ctx.step_info(n)["balanced"]
    a1 = torch.zeros((n, 4), device="cuda")
    ctx.accel_range(x, a1, 0, n, 0, n)
    ctx.sync()
    junk = torch.zeros((3000, 4), device="cuda")
    ctx.accel_range(x, junk, 0, 3000, 2000, 9000)              # a rectangle: one-sided kernel, writes slabs over the inboxes
    big = torch.from_numpy(nb.engine.seeded_bodies(30000, 1, 6)).cuda()
    abig = torch.zeros((30000, 4), device="cuda")
    ctx.accel_range(big, abig, 0, 30000, 0, 30000)              # another balanced layout (and a larger workspace)
    ctx.set_symmetric_runs(0)
    ctx.accel_range(big, abig, 0, 30000, 0, 30000)              # block pairs over the same workspace
    ctx.set_symmetric_runs(-1)
    a2 = torch.zeros((n, 4), device="cuda")
    ctx.accel_range(x, a2, 0, n, 0, n)
    ctx.sync()
    assert torch.equal(a1, a2)
    truth = oracle.accel_range(x0, 0, 1024, 0, n, eps2=0.002, f64acc=True)
    assert np.abs(a2.cpu().numpy()[:1024] - truth)[:, :3].max() / np.abs(truth[:, :3]).max() <= 1e-5


# ---- the fused small-N step: force + integrate in ONE launch, positions alternating between two arrays ---------------------

@pytest.mark.parametrize("n,init,steps", [(1, 0, 3), (2, 0, 2), (63, 1, 3), (64, 0, 1), (129, 1, 4), (1000, 0, 5), (1024, 1, 100), (2048, 0, 1),
                                          (3001, 1, 1), (4096, 0, 1), (5000, 1, 2), (8192, 0, 1), (8192, 0, 3), (8192, 1, 10), (7777, 1, 3),
                                          (8192, 1, 1)])
def test_fused_step_vs_oracle(nb, oracle, n, init, steps):
    """nbody_step through the fused kernel (FAST, n <= 8192): every workgroup shape the size rule picks, sizes that are not
    multiples of anything, odd and even step counts (the result must end in the caller's array either way), against the Jacobi
    oracle: accelerations <= 1e-5 of max|a|, positions within the fast tolerances, mass and w lanes preserved, bitwise
    run-to-run reproducibility; and equal (to tolerance) to the two-kernel path it replaces."""
    x0 = nb.engine.seeded_bodies(n, init, 31)
    dt = 0.1 if init == 0 else 0.01
    scale = 1e5 if init == 0 else 1.0
    sim = nb.engine.Simulation(x0, dt=dt, eps2=0.002)
    assert sim.ctx.step_info(n)["fused"]
    sim.run(steps)
    x, v, a = sim.state()
    xo, vo, ao = x0.copy(), np.zeros_like(x0), np.zeros_like(x0)
    oracle.step_jacobi(xo, ao, vo, dt=dt, eps2=0.002, steps=steps)
    amax = max(np.abs(ao[:, :3]).max(), 1e-30)
    tol_x = 1e-6 if steps <= 5 else 1e-5
    assert np.abs(x - xo)[:, :3].max() / scale <= tol_x
    if steps == 1:     # (later steps evaluate the force at positions that already differ in the last bits: close pairs amplify that)
        assert np.abs(a - ao)[:, :3].max() / amax <= 1e-5
    assert np.array_equal(x[:, 3], x0[:, 3]) and np.all(a[:, 3] == 0) and np.all(v[:, 3] == 0)
    again = nb.engine.Simulation(x0, dt=dt, eps2=0.002)
    again.run(steps)
    x2, v2, a2 = again.state()
    assert np.array_equal(x, x2) and np.array_equal(v, v2) and np.array_equal(a, a2)
    plain = nb.engine.Simulation(x0, dt=dt, eps2=0.002)
    plain.ctx.set_fused(0)
    assert not plain.ctx.step_info(n)["fused"]
    plain.run(steps)
    xp, vp, ap = plain.state()
    assert np.abs(x - xp)[:, :3].max() / scale <= tol_x


def test_fused_step_split_calls_and_the_simulate_boundary(nb, oracle):
    """Seven steps as 1 + 2 + 4 separate calls equal seven in one call bit for bit (the spare array never leaks into the result);
    simulate() — one synchronous step per call, the reference's boundary — takes the same path; mode 1 forces the fused step at a
    size where FAST would not choose it."""
    n = 3000
    x0 = nb.engine.seeded_bodies(n, 1, 8)
    a_sim = nb.engine.Simulation(x0, dt=0.01, eps2=0.002)
    a_sim.run(7)
    b_sim = nb.engine.Simulation(x0, dt=0.01, eps2=0.002)
    for k in (1, 2, 4):
        b_sim.run(k)
    for p, q in zip(a_sim.state(), b_sim.state()):
        assert np.array_equal(p, q)
    dev = torch.device("cuda", 0)
    x = torch.from_numpy(nb.engine.seeded_bodies(n, 0, 9)).to(dev)
    v, a = torch.zeros_like(x), torch.zeros_like(x)
    x_start = x.cpu().numpy()
    for _ in range(3):
        nb.engine.simulate(x, a, v)                               # DT 0.1 / EPS2 0.002, default context, synchronous
    xo, vo, ao = x_start.copy(), np.zeros_like(x_start), np.zeros_like(x_start)
    oracle.step_jacobi(xo, ao, vo, dt=0.1, eps2=0.002, steps=3)
    assert np.abs(x.cpu().numpy() - xo)[:, :3].max() / 1e5 <= 1e-6
    n2 = 20000
    big = nb.engine.Simulation(nb.engine.seeded_bodies(n2, 1, 4), dt=0.01, eps2=0.002)
    assert big.ctx.step_info(n2)["balanced"]
    big.ctx.set_fused(1)
    assert big.ctx.step_info(n2)["fused"]
    big.run(1)
    truth = oracle.accel_range(nb.engine.seeded_bodies(n2, 1, 4), 0, 512, 0, n2, eps2=0.002, f64acc=True)
    assert np.abs(big.state()[2][:512] - truth)[:, :3].max() / np.abs(truth[:, :3]).max() <= 1e-5


@pytest.mark.parametrize("n,init", [(1, 0), (2, 1), (63, 0), (129, 1), (1000, 0), (2048, 1), (3001, 0), (4096, 1), (6144, 0), (8000, 1), (8192, 0)])
def test_fused_step_in_place_is_bit_identical_to_the_two_array_step(nb, n, init):
    """nbk::step_fused<.., INPLACE>: the caller's position array read and written by ONE launch (a non-blocking protocol between the
    workgroups decides per wave whether everybody has read; waves that cannot know write to the spare array and the last workgroup
    moves their positions home). Same arithmetic, so the same BITS as the two-array kernel + copy-back — in every mode: the default
    (the odd last step of a call in place), every step in place, and every wave forced down the fall-back path (mode 2: the repair
    by the last workgroup carries the whole result); odd and even step counts, steps split over calls."""
    x0 = nb.engine.seeded_bodies(n, init, 4242)
    dt = 0.1 if init == 0 else 0.01
    waves = (n + 1) // 2 if n <= 8192 else 0

    def run(mode, calls):
        sim = nb.engine.Simulation(x0, dt=dt, eps2=0.002)
        assert sim.ctx.step_info(n)["fused"]
        sim.ctx.set_fused_inplace(mode)
        for k in calls:
            sim.run(k)
        return sim.state(), sim.ctx.fused_inplace_fallbacks()

    want, fb0 = run(0, (5,))
    assert fb0 == 0
    for mode, calls in ((-1, (5,)), (-1, (1, 1, 3)), (-1, (2, 3)), (1, (5,)), (1, (1, 4)), (2, (5,)), (2, (3, 1, 1))):
        got, fb = run(mode, calls)
        for p, q in zip(want, got):
            assert np.array_equal(p, q), (mode, calls)
        if mode == 2:      # every wave of every in-place launch took the fall-back path and was repaired
            assert fb == waves * 5, (fb, waves)
    want4, _ = run(0, (4,))
    for mode in (-1, 1, 2):
        got, _ = run(mode, (4,))
        for p, q in zip(want4, got):
            assert np.array_equal(p, q), mode


def test_simulate_takes_the_in_place_step_and_its_host_word(nb, oracle):
    """simulate() — one synchronous step per call, the reference's loop (main.cpp:146-156) — runs the in-place fused step and
    waits on the host-mapped word the launch writes. The device arrays must be complete when the call returns: they are read back
    at once, by a plain copy, after every call, and compared bit for bit with the queued two-array run."""
    dev = torch.device("cuda", 0)
    for n in (8192, 3000, 100):
        x0 = nb.engine.seeded_bodies(n, 0, 9)
        ref = nb.engine.Simulation(x0)                                   # DT 0.1 / EPS2 0.002: the defaults simulate() runs with
        ref.ctx.set_fused_inplace(0)
        x = torch.from_numpy(x0).to(dev)
        v, a = torch.zeros_like(x), torch.zeros_like(x)
        for k in range(1, 8):
            nb.engine.simulate(x, a, v)
            got = (x.cpu().numpy(), v.cpu().numpy(), a.cpu().numpy())    # straight after the call returns
            if k == 1:     # nothing is measured unless the caller opts in (NBODY_AUTOTUNE): the built-in decomposition, on every machine
                assert nb.engine.simulate_autotuned(n)["choice"] == -1
            ref.run(1)
            for p, q in zip(ref.state(), got):
                assert np.array_equal(p, q), (n, k)
    xo, vo, ao = x0.copy(), np.zeros_like(x0), np.zeros_like(x0)
    oracle.step_jacobi(xo, ao, vo, dt=0.1, eps2=0.002, steps=7)
    assert np.abs(got[0] - xo)[:, :3].max() / 1e5 <= 1e-6


def test_simulate_results_read_from_another_stream_at_once(nb):
    """the arrays are complete when simulate() returns, "whichever way the caller reads them next". Here the next
    reader is a kernel on ANOTHER, non-blocking stream (nothing orders it behind the library's stream but the call having returned):
    a device-side copy of all three arrays, taken right after every call, must carry the bits of the same steps run queued — at the
    fused in-place step's sizes (the launch itself tells the host, while it is still winding down) and above (host word behind the step)."""
    dev = torch.device("cuda", 0)
    side = torch.cuda.Stream(device=dev)
    for n in (8192, 5000, 20000):
        x0 = nb.engine.seeded_bodies(n, 0, 17)
        x = torch.from_numpy(x0).to(dev)
        v, a = torch.zeros_like(x), torch.zeros_like(x)
        ref = nb.engine.Simulation(x0)
        torch.cuda.synchronize(dev)
        for k in range(6):
            nb.engine.simulate(x, a, v)
            with torch.cuda.stream(side):                      # no event, no synchronisation with the library's stream
                cx, cv, ca = x.clone(), v.clone(), a.clone()
            side.synchronize()
            ref.run(1)
            for p, q in zip(ref.state(), (cx.cpu().numpy(), cv.cpu().numpy(), ca.cpu().numpy())):
                assert np.array_equal(p, q), (n, k)


def test_simulate_on_host_mapped_arrays_read_by_the_cpu_at_once(nb):
    """simulate() is synchronous (kernel.cu:644): whatever memory the caller's arrays live in, they are complete when the call
    returns. Here all three arrays are HOST-MAPPED (nbody_malloc_host) or MANAGED (hipMallocManaged) and the CPU reads them directly —
    no copy, no HIP call in between — right after every call: at N <= 8192 the in-place fused step ends with system-scope stores drained before its word;
    above, the library sees that the arrays are not device memory and pays a stream synchronisation instead of its host word. The
    bits must be those of the same steps on device arrays."""
    import ctypes as C
    lib = nb.load()
    hip = C.CDLL("libamdhip64.so")            # (already in the process: the library links it) for hipMallocManaged, which the C-ABI does not wrap
    hip.hipMallocManaged.argtypes = [C.POINTER(C.c_void_p), C.c_size_t, C.c_uint]
    hip.hipFree.argtypes = [C.c_void_p]
    for n, kind in ((8192, "host"), (12288, "host"), (50000, "host"), (8192, "managed"), (20000, "managed")):
        x0 = nb.engine.seeded_bodies(n, 0, 31)
        ptrs, views = [], []
        try:
            for _ in range(3):
                h = C.c_void_p()
                if kind == "host":
                    nb.engine.check(lib.nbody_malloc_host(C.byref(h), 16 * n))
                elif hip.hipMallocManaged(C.byref(h), 16 * n, 1) != 0:      # hipMemAttachGlobal
                    pytest.skip("hipMallocManaged is not available on this box")
                ptrs.append(h)
                views.append(np.ctypeslib.as_array(C.cast(h, C.POINTER(C.c_float)), shape=(n, 4)))
            hx, ha, hv = views
            hx[:] = x0
            ha[:] = 0
            hv[:] = 0
            ref = nb.engine.Simulation(x0)
            for k in range(5):
                nb.engine.check(lib.nbody_simulate(ptrs[0], ptrs[1], ptrs[2], n))
                got = (hx.copy(), hv.copy(), ha.copy())        # the CPU's own loads, straight after the call returns
                ref.run(1)
                for p, q in zip(ref.state(), got):
                    assert np.array_equal(p, q), (n, kind, k)
        finally:
            for h in ptrs:
                lib.nbody_free_host(h) if kind == "host" else hip.hipFree(h)


def test_simulate_measures_nothing_unless_asked(nb, monkeypatch):
    """The reference's loop never asked for tuning (main.cpp:146-156): by default simulate() measures nothing, even at a switch-over
    size, and computes the bits of the built-in decomposition."""
    monkeypatch.delenv("NBODY_AUTOTUNE", raising=False)
    dev = torch.device("cuda", 0)
    for n in (8190, 44000):
        x0 = nb.engine.seeded_bodies(n, 0, 77)
        x = torch.from_numpy(x0).to(dev)
        v, a = torch.zeros_like(x), torch.zeros_like(x)
        nb.engine.simulate(x, a, v)
        nb.engine.simulate(x, a, v)
        assert nb.engine.simulate_autotuned(n) == {"choice": -1, "us_builtin": 0.0, "us_best": 0.0}
        ref = nb.engine.Simulation(x0)
        ref.run(2)
        for p, q in zip(ref.state(), (x.cpu().numpy(), v.cpu().numpy(), a.cpu().numpy())):
            assert np.array_equal(p, q), n


def test_simulate_measures_the_decompositions_near_a_switch_over_size(nb, oracle, monkeypatch):
    """nbody_simulate() on the default context WITH NBODY_AUTOTUNE=1 (opt-in): the first call with n within a quarter of a built-in
    switch-over size (8192, 45056, 160000) times the decompositions on scratch copies and keeps the built-in one unless another is
    more than 3 % faster; sizes elsewhere are not measured; the caller's bodies are only read by the measurement; what simulate()
    then computes is bit-identical to a context forced onto the reported decomposition."""
    monkeypatch.setenv("NBODY_AUTOTUNE", "1")
    dev = torch.device("cuda", 0)
    for n, near in ((9000, True), (3000, False), (40000, True), (70000, False)):
        x0 = nb.engine.seeded_bodies(n, 0, 123)
        x = torch.from_numpy(x0).to(dev)
        v, a = torch.zeros_like(x), torch.zeros_like(x)
        nb.engine.simulate(x, a, v)
        r = nb.engine.simulate_autotuned(n)
        if near:
            assert r["choice"] in (0, 1, 24, 28, 210, 3, 4) and r["us_builtin"] > 0 and 0 < r["us_best"] <= r["us_builtin"], (n, r)
            if r["choice"] != 0:
                assert r["us_best"] < 0.97 * r["us_builtin"], r      # the built-in choice is only overridden by a clear win
        else:
            assert r["choice"] == -1 and r["us_best"] == 0.0, (n, r)
        nb.engine.simulate(x, a, v)
        ref = nb.engine.Simulation(x0)                                # defaults of constants.h, like simulate()
        nb.engine.force_choice(ref.ctx, r["choice"])
        ref.run(2)
        for p, q in zip(ref.state(), (x.cpu().numpy(), v.cpu().numpy(), a.cpu().numpy())):
            assert np.array_equal(p, q), (n, r)
        xo, vo, ao = x0.copy(), np.zeros_like(x0), np.zeros_like(x0)
        if n <= 9000:
            oracle.step_jacobi(xo, ao, vo, dt=0.1, eps2=0.002, steps=2)
            assert np.abs(x.cpu().numpy() - xo)[:, :3].max() / 1e5 <= 1e-6


@pytest.mark.parametrize("n,choice", [(9000, 1), (9000, 24), (9000, 28), (9000, 4), (40000, 210), (40000, 3), (40000, 4), (5000, 24)])
def test_simulate_applies_a_pinned_decomposition_bit_for_bit(nb, n, choice):
    """Whatever a measurement may find on some other machine, simulate() applies it exactly: with the decomposition for n pinned
    (nbody_ctx_set_autotuned) two simulate() calls give the bits of a context forced onto that decomposition through the knobs, and
    forgetting the pin restores the built-in choice."""
    dev = torch.device("cuda", 0)
    x0 = nb.engine.seeded_bodies(n, 0, 321)
    try:
        nb.engine.simulate_pin(n, choice)
        assert nb.engine.simulate_autotuned(n)["choice"] == choice
        x = torch.from_numpy(x0).to(dev)
        v, a = torch.zeros_like(x), torch.zeros_like(x)
        nb.engine.simulate(x, a, v)
        nb.engine.simulate(x, a, v)
        ref = nb.engine.Simulation(x0)
        nb.engine.force_choice(ref.ctx, choice)
        info = ref.ctx.step_info(n)
        assert {1: info["fused"], 24: info["balanced"], 28: info["balanced"], 210: info["balanced"], 3: info["runs"],
                4: not (info["fused"] or info["balanced"] or info["runs"])}[choice], info
        ref.run(2)
        for p, q in zip(ref.state(), (x.cpu().numpy(), v.cpu().numpy(), a.cpu().numpy())):
            assert np.array_equal(p, q)
    finally:
        nb.engine.simulate_pin(n, 0 if n == 5000 else -1)
    with pytest.raises(nb.NBodyError):
        nb.engine.simulate_pin(n, 7)


def test_autotune_measures_and_sets_the_knobs(nb, oracle):
    """nbody_ctx_autotune: times the decompositions that apply to whole steps of n bodies on the device at hand (scratch copies,
    dt = 0: the caller's array is only read) and leaves the context on the fastest. On MI355X that reproduces the built-in choice at
    the sizes it was measured at; results afterwards stay within the fast tolerances whatever it picked."""
    for n, expect in ((8192, (1,)), (16384, (24, 28, 210)), (2048, (1,))):
        x0 = nb.engine.seeded_bodies(n, 1, 77)
        sim = nb.engine.Simulation(x0, dt=0.01, eps2=0.002)
        before = sim.x.clone()
        r = sim.ctx.autotune(sim.x, steps_per_trial=40)
        assert torch.equal(sim.x, before)                                   # the caller's bodies were only read
        assert r["choice"] in expect, (n, r)
        assert 1.0 < r["us_per_step"] < (35.0 if n <= 8192 else 90.0), (n, r)
        info = sim.ctx.step_info(n)
        assert info["fused"] == (r["choice"] == 1) and info["balanced"] == (r["choice"] >= 24)
        sim.run(1)
        truth = oracle.accel_range(x0, 0, 512, 0, n, eps2=0.002, f64acc=True)
        assert np.abs(sim.state()[2][:512] - truth)[:, :3].max() / np.abs(truth[:, :3]).max() <= 1e-5
    ctx = nb.engine.Context(kernel=nb.KERNEL_STRICT)
    with pytest.raises(nb.NBodyError):
        ctx.autotune(torch.zeros((64, 4), device="cuda"))


# ---- equal-mass path of the symmetric kernels (decided on the device, per launch) -----------------------------------

def _accel_all(nb, x0, eq_mode, kernel=None):
    ctx = nb.engine.Context(kernel=nb.KERNEL_FAST if kernel is None else kernel)
    ctx.set_equal_mass(eq_mode)
    n = len(x0)
    x = torch.from_numpy(x0).cuda()
    a = torch.full((n, 4), float("nan"), device="cuda")
    ctx.accel_range(x, a, 0, n, 0, n)
    ctx.sync()
    return a.cpu().numpy(), ctx


@pytest.mark.parametrize("n,kind", [(20000, "balanced"), (65536, "runs"), (100003, "runs"), (200000, "symmetric"), (262144, "symmetric")])
def test_equal_mass_path_vs_general_path(nb, oracle, n, kind):
    """Plummer bodies all carry the mass 1/N: the device-side scan finds them uniform and the symmetric kernel of every
    decomposition (balanced runs, unit runs, block pairs; ragged sizes with padding lanes) accumulates sum w r and applies the
    common mass once. Against the general path (switched off on a second context): a rounding-level difference; against the
    fp64-accumulated CPU sums at both ends of the array: the usual 1e-5; exact mass linearity; bitwise repeat."""
    x0 = nb.engine.seeded_bodies(n, 1, 2024)
    assert np.all(x0[:, 3] == x0[0, 3])
    a_eq, ctx = _accel_all(nb, x0, 1)
    info = ctx.step_info(n)
    assert info[kind] and (kind == "symmetric" or info["symmetric"]), info
    verdict = ctx.equal_mass_verdict()
    assert verdict["scanned"] and verdict["uniform"] and verdict["mass"] == x0[0, 3]
    a_gen, ctx0 = _accel_all(nb, x0, 0)
    assert not ctx0.equal_mass_verdict()["scanned"]
    scale = np.abs(a_gen[:, :3]).max()
    assert np.isfinite(a_eq).all() and np.all(a_eq[:, 3] == 0)
    assert np.abs(a_eq - a_gen)[:, :3].max() <= 2e-6 * scale
    # the other path did run: different rounding — unless the common mass is a power of two (N = 65536, 262144: 1/N), where
    # scaling is exact and m0 * sum(w r) == sum((m0 w) r) bit for bit
    assert np.array_equal(a_eq, a_gen) == (np.frexp(x0[0, 3])[0] == 0.5)
    for i0 in (0, n - 200):
        truth = oracle.accel_range(x0, i0, i0 + 200, 0, n, eps2=0.002, f64acc=True)
        assert np.abs(a_eq[i0:i0 + 200] - truth)[:, :3].max() / np.abs(truth[:, :3]).max() <= 1e-5
    again, _ = _accel_all(nb, x0, 1)
    assert np.array_equal(again, a_eq)
    x4 = x0.copy(); x4[:, 3] *= 4
    a4, _ = _accel_all(nb, x4, 1)
    assert np.array_equal(a4[:, :3], 4 * a_eq[:, :3])


@pytest.mark.parametrize("n", [20000, 65536, 200000])
def test_one_different_body_takes_the_general_path_bit_for_bit(nb, oracle, n):
    """The scan must see EVERY body: one mass that differs in its last bit (first, middle or last body of the array), a
    coordinate beyond 1e15 or a NaN send the launch down the general path — the same bits as with the equal-mass path
    switched off."""
    x0 = nb.engine.seeded_bodies(n, 1, 7)
    for where in (0, n // 2 + 1, n - 1):
        x1 = x0.copy()
        x1[where, 3] = np.nextafter(x1[where, 3], np.float32(1.0))
        a_on, ctx = _accel_all(nb, x1, 1)
        v = ctx.equal_mass_verdict()
        assert v["scanned"] and not v["uniform"], (where, v)
        a_off, _ = _accel_all(nb, x1, 0)
        assert np.array_equal(a_on, a_off), where
    x2 = x0.copy()
    x2[n // 3, 0] = 3e15                                         # uniform masses, but a body the far-away padding is not far from
    a_on, ctx = _accel_all(nb, x2, 1)
    assert not ctx.equal_mass_verdict()["uniform"]
    a_off, _ = _accel_all(nb, x2, 0)
    assert np.array_equal(a_on, a_off)
    x3 = x0.copy()
    x3[5, 1] = np.nan
    a_on, ctx = _accel_all(nb, x3, 1)
    assert not ctx.equal_mass_verdict()["uniform"]
    a_off, _ = _accel_all(nb, x3, 0)
    assert np.array_equal(a_on, a_off, equal_nan=True)


def test_equal_mass_cross_launch_and_whole_steps(nb, oracle):
    """nbody_accel_cross (two disjoint ranges, the source run wrapping round the array) on
    the equal-mass path: both sides' sums against the CPU; a run that contains one heavier body: the general path, same bits as
    with the path switched off. Whole steps (N = 65536, 5 steps): positions of the two paths agree to 1e-6 of the scale radius,
    and the reference's own initial conditions (random masses) never take the path."""
    n = 60000
    x0 = nb.engine.seeded_bodies(n, 1, 99)
    x = torch.from_numpy(x0).cuda()
    i0, i1, j0, cnt = 10000, 30000, 50000, 19000                 # sources 50000..59999, 0..8999
    src = np.r_[np.arange(j0, n), np.arange(0, j0 + cnt - n)]
    xs = np.concatenate([x0[i0:i1], x0[src]])
    ti = oracle.accel_range(xs, 0, 256, i1 - i0, len(xs), eps2=0.002, f64acc=True)
    tj = oracle.accel_range(xs, i1 - i0 + cnt - 256, i1 - i0 + cnt, 0, i1 - i0, eps2=0.002, f64acc=True)
    for shape in ((0, 0), (4, 10), (4, 8)):                      # (4,10): the rectangular-only build of the kernel; (4,8): the general one
        res = {}
        for mode in (1, 0):
            ctx = nb.engine.Context()
            ctx.set_equal_mass(mode)
            ctx.set_symmetric_shape(*shape)
            ai = torch.zeros((i1 - i0, 4), device="cuda")
            aj = torch.zeros((cnt, 4), device="cuda")
            ctx.accel_cross(x, ai, i0, i1, False, j0, cnt, aj)
            ctx.sync()
            res[mode] = (ai.cpu().numpy(), aj.cpu().numpy())
            if mode == 1:
                assert ctx.equal_mass_verdict()["uniform"]
        for mode in (1, 0):
            assert np.abs(res[mode][0][:256] - ti)[:, :3].max() / np.abs(ti[:, :3]).max() <= 1e-5, shape
            assert np.abs(res[mode][1][-256:] - tj)[:, :3].max() / np.abs(tj[:, :3]).max() <= 1e-5, shape
        assert not np.array_equal(res[1][0], res[0][0])         # 1/60000 is not a power of two: the roundings differ
    x1 = x0.copy(); x1[3, 3] *= 2                                # a heavier body inside the wrapped part of the source run
    xh = torch.from_numpy(x1).cuda()
    out = {}
    for mode in (1, 0):
        ctx = nb.engine.Context()
        ctx.set_equal_mass(mode)
        ai = torch.zeros((i1 - i0, 4), device="cuda")
        aj = torch.zeros((cnt, 4), device="cuda")
        ctx.accel_cross(xh, ai, i0, i1, False, j0, cnt, aj)
        ctx.sync()
        out[mode] = (ai.cpu().numpy(), aj.cpu().numpy())
        if mode == 1:
            assert not ctx.equal_mass_verdict()["uniform"]
    assert np.array_equal(out[1][0], out[0][0]) and np.array_equal(out[1][1], out[0][1])
    # whole steps
    n = 65536
    x0 = nb.engine.seeded_bodies(n, 1, 5)
    fin = {}
    for mode in (1, 0):
        sim = nb.engine.Simulation(x0, dt=0.01, eps2=0.002)
        sim.ctx.set_equal_mass(mode)
        sim.run(5)
        fin[mode] = sim.state()
        assert sim.ctx.equal_mass_verdict()["uniform"] == (mode == 1)
    assert np.abs(fin[1][0] - fin[0][0])[:, :3].max() <= 1e-6
    assert np.abs(fin[1][2] - fin[0][2])[:, :3].max() <= 3e-6 * np.abs(fin[0][2][:, :3]).max()
    xr = nb.engine.seeded_bodies(n, 0, 5)                        # the reference's kind of initial conditions: random masses
    sim = nb.engine.Simulation(xr, dt=0.1, eps2=0.002)
    sim.run(2)
    v = sim.ctx.equal_mass_verdict()
    assert v["scanned"] and not v["uniform"]


def test_equal_mass_zero_and_negative_common_mass(nb, oracle):
    """A common mass of zero gives exact zeros, a negative one the mirrored field; nothing non-finite appears."""
    n = 30000
    x0 = nb.engine.seeded_bodies(n, 1, 3)
    ref, _ = _accel_all(nb, x0, 1)
    xz = x0.copy(); xz[:, 3] = 0.0
    az, ctx = _accel_all(nb, xz, 1)
    assert ctx.equal_mass_verdict()["uniform"] and np.all(az == 0)
    xn = x0.copy(); xn[:, 3] *= -1
    an, _ = _accel_all(nb, xn, 1)
    assert np.array_equal(an[:, :3], -ref[:, :3])


def test_equal_mass_path_in_double(nb, oracle):
    """The fp64 step on the equal-mass path (same scan, padding at 1e150): N = 30001 (ragged) and the general path on the same
    bodies agree to rounding; both within 1e-12 of the checker's all-double sums; one different mass switches the path off
    (bit-identical to the path switched off)."""
    n = 30001
    x0 = nb.engine.seeded_bodies(n, 1, 17).astype(np.float64)
    x0[:, 3] = 1.0 / n                                            # a double that is not a rounded float
    out = {}
    for mode in (1, 0):
        ctx = nb.engine.Context()
        ctx.set_equal_mass(mode)
        x = torch.from_numpy(x0).cuda()
        v = torch.zeros_like(x)
        a = torch.zeros_like(x)
        ctx.step_f64(x, a, v, dt=0.01, eps2=0.002, steps=1)
        ctx.sync()
        assert ctx.step_info_f64(n)["symmetric"]
        ver = ctx.equal_mass_verdict()
        assert ver["scanned"] == (mode == 1) and ver["uniform"] == (mode == 1)
        out[mode] = a.cpu().numpy()
    scale = np.abs(out[0][:, :3]).max()
    assert np.abs(out[1] - out[0])[:, :3].max() <= 1e-14 * scale and not np.array_equal(out[1], out[0])
    for i0 in (0, n - 128):
        want = oracle.accel_range_f64(x0, i0, i0 + 128, 0, n, eps2=0.002)
        for mode in (1, 0):
            assert np.abs(out[mode][i0:i0 + 128] - want)[:, :3].max() / np.abs(want[:, :3]).max() <= 1e-12
    x1 = x0.copy(); x1[n - 1, 3] *= 1.0 + 2.0 ** -40
    res = {}
    for mode in (1, 0):
        ctx = nb.engine.Context()
        ctx.set_equal_mass(mode)
        x = torch.from_numpy(x1).cuda()
        v = torch.zeros_like(x)
        a = torch.zeros_like(x)
        ctx.step_f64(x, a, v, dt=0.01, eps2=0.002, steps=1)
        ctx.sync()
        if mode == 1:
            assert not ctx.equal_mass_verdict()["uniform"]
        res[mode] = a.cpu().numpy()
    assert np.array_equal(res[1], res[0])


def test_equal_mass_api_corners(nb, oracle):
    """Mode range, the verdict before any scan and below the scan's minimum size, a stream change between launches of the balanced
    runs (the inbox clear is re-issued on the new stream), and the verdict after a run of the fused small-N step (no scan there)."""
    ctx = nb.engine.Context()
    for bad in (-2, 2, 7):
        with pytest.raises(nb.NBodyError):
            ctx.set_equal_mass(bad)
    assert ctx.equal_mass_verdict() == {"scanned": False, "uniform": False, "mass": 0.0}
    n = 3000                                                     # below 4096 bodies no scan is launched, whatever the mode
    x0 = nb.engine.seeded_bodies(n, 1, 1)
    sim = nb.engine.Simulation(x0, dt=0.01, eps2=0.002, kernel=nb.KERNEL_SYMMETRIC)
    sim.ctx.set_equal_mass(1)
    sim.run(2)
    assert not sim.ctx.equal_mass_verdict()["scanned"]
    for n, scans in ((20000, False), (32768, True)):             # the automatic mode leaves launches below 32768 bodies alone
        sim = nb.engine.Simulation(nb.engine.seeded_bodies(n, 1, 1), dt=0.01, eps2=0.002)
        sim.run(1)
        assert sim.ctx.step_info(n)["symmetric"] and sim.ctx.equal_mass_verdict()["scanned"] == scans, n
    sim = nb.engine.Simulation(nb.engine.seeded_bodies(8192, 1, 1), dt=0.01, eps2=0.002)   # fused step: one-sided arithmetic, no scan
    sim.run(3)
    assert sim.ctx.step_info(8192)["fused"] and not sim.ctx.equal_mass_verdict()["scanned"]
    n = 12000                                                    # balanced runs on two streams in turn
    x0 = nb.engine.seeded_bodies(n, 1, 2)
    x = torch.from_numpy(x0).cuda()
    ref, _ = _accel_all(nb, x0, 1)
    ctx = nb.engine.Context()
    ctx.set_equal_mass(1)
    assert ctx.step_info(n)["balanced"]
    a = torch.zeros((n, 4), device="cuda")
    ctx.accel_range(x, a, 0, n, 0, n)
    ctx.sync()
    side = torch.cuda.Stream()
    ctx.set_stream(side)
    a2 = torch.zeros((n, 4), device="cuda")
    side.wait_stream(torch.cuda.current_stream())
    ctx.accel_range(x, a2, 0, n, 0, n)
    ctx.sync()
    ctx.set_stream(None)
    assert np.array_equal(a.cpu().numpy(), ref) and np.array_equal(a2.cpu().numpy(), ref)
    assert ctx.equal_mass_verdict()["uniform"]
