"""The measurement hooks of the library: event timing of the force launches and the clock stamps around them
(nbody_ctx_timing(ctx, 2) / nbody_ctx_clock_read) that bench.py's roofline block reports — the reference measures nothing
(TestProject/main.cpp:142-160: two printf), so these are the build's own contract."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_clock_stamps_bracket_every_timed_force_launch(nb):
    """Shader cycles and 100-MHz ticks of all eight XCDs around each force launch: every launch gets its record, cycles / sclk
    reproduces the event time, the clock is a plausible MI355X shader clock, and stepping with stamps gives the same bits as without."""
    n, steps = 65536, 6
    x0 = nb.engine.seeded_bodies(n, 1, 12345)
    plain = nb.engine.Simulation(x0, dt=0.01)
    plain.ctx.set_equal_mass(0)
    plain.run(steps)
    sim2 = nb.engine.Simulation(x0, dt=0.01)
    sim2.ctx.set_equal_mass(0)
    sim2.ctx.timing(True, clock=True)
    sim2.run(steps, sync=False)
    ms, launches = sim2.ctx.timing_read()
    ck = sim2.ctx.clock_read()
    sim2.ctx.timing(False)
    assert launches == steps and ck["launches"] == steps, (launches, ck)
    assert ck["xcds"] == 8 and ck["unpaired"] <= steps, ck                                  # every XCD answered (a launch may miss one now and then)
    assert 800.0 < ck["sclk_mhz"] < 2600.0 and ck["sclk_mhz_min_xcd"] <= ck["sclk_mhz"] <= ck["sclk_mhz_max_xcd"], ck
    assert ck["sclk_mhz_max_xcd"] - ck["sclk_mhz_min_xcd"] < 0.06 * ck["sclk_mhz"], ck      # the XCDs run 1.5-2 % apart (tools/clock_probe.hip)
    event_ms = ms / launches
    by_clock_ms = ck["ticks_per_launch"] * 1e-5                                             # 100-MHz ticks -> ms
    # the stamps sit OUTSIDE the event pair: the clock's interval is the event's plus two launch boundaries (a few microseconds)
    assert event_ms <= by_clock_ms * 1.002 and by_clock_ms - event_ms < 0.05, (event_ms, by_clock_ms)
    assert abs(ck["cycles_per_launch"] / (ck["sclk_mhz"] * 1e3) - by_clock_ms) < 1e-6 * by_clock_ms + 1e-9
    assert ck["cycles_per_launch_min"] <= ck["cycles_per_launch"] <= ck["cycles_per_launch_max"]
    # a second read without launches in between is empty
    assert sim2.ctx.clock_read()["launches"] == 0
    for p, q in zip(plain.state(), sim2.state()):
        assert np.array_equal(p, q)


def test_clock_stamps_survive_more_launches_than_their_buffer(nb):
    """4096 stamped launches are kept between two reads; later launches go unstamped (events keep counting) and nothing is overrun."""
    n, steps = 4096, 4200
    sim = nb.engine.Simulation(nb.engine.seeded_bodies(n, 1, 7), dt=0.01)
    sim.ctx.timing(True, clock=True)
    sim.run(steps, sync=False)
    ms, launches = sim.ctx.timing_read()
    ck = sim.ctx.clock_read()
    sim.ctx.timing(False)
    assert launches == steps and ck["launches"] == 4096 and ck["unpaired"] < 4096, (launches, ck)
    assert 800.0 < ck["sclk_mhz"] < 2600.0
