"""ISA-level guard on the one intra-launch, cross-workgroup protocol of the library: the in-place fused step.

`step_fused<.., INPLACE = true>` (csrc/nbody_kernels.hip.h) lets the workgroup that completes the `finished` count write a host-mapped
word on which `nbody_simulate` returns — the synchrony the reference gives its caller with cudaDeviceSynchronize
(/root/reference/TestProject/kernel.cu:644; validation.cpp:77-81 copies the arrays back right after). That is only sound if EVERY
storing wave has drained its result stores (s_waitcnt vmcnt(0): on gfx9-family parts stores count in vmcnt, and a system-scope
`sc0 sc1` store is acknowledged by memory) BEFORE its workgroup's barrier and count-out (MI355X_MICROARCH.md, "every storing wave's
s_waitcnt vmcnt(0), the workgroup's barrier, then the flag/counter"). The C++ source cannot promise that — the compiler places waits
— and no amount of running shows the hole (it is a sub-microsecond window), so this test reads the instructions of the SHIPPED
library: it unbundles libnbody_hip.so's gfx950 code object and disassembles it. No GPU needed.
"""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM = "/opt/rocm/lib/llvm/bin"
LIB = os.path.join(ROOT, "n-bodysimulation_amd", "libnbody_hip.so")

INSN = re.compile(r"^\s+(\S+)\s*(.*?)\s*//\s*([0-9A-Fa-f]+):")
TARGET = re.compile(r"<[^>]*\+0x([0-9a-fA-F]+)>\s*$")


def disassemble(lib, workdir):
    """{mangled kernel name: [(address, mnemonic, operands, branch target address or None)]} of every gfx950 object bundled in lib"""
    local = os.path.join(workdir, os.path.basename(lib))
    shutil.copy(lib, local)                      # llvm-objdump --offloading writes the extracted objects beside its input
    subprocess.run([os.path.join(LLVM, "llvm-objdump"), "--offloading", local], check=True, capture_output=True, cwd=workdir)
    objs = sorted(f for f in os.listdir(workdir) if "gfx950" in f)
    assert objs, "no gfx950 code object in " + lib
    funcs = {}
    for o in objs:
        txt = subprocess.run([os.path.join(LLVM, "llvm-objdump"), "-d", "--mcpu=gfx950", os.path.join(workdir, o)],
                             check=True, capture_output=True, text=True).stdout
        cur, base = None, 0
        for ln in txt.splitlines():
            m = re.match(r"^([0-9a-f]+) <(\S+)>:$", ln)
            if m:
                base, cur = int(m.group(1), 16), m.group(2)
                funcs[cur] = []
                continue
            m = INSN.match(ln)
            if m and cur:
                t = TARGET.search(ln) if m.group(1).startswith(("s_cbranch", "s_branch")) else None
                funcs[cur].append((int(m.group(3), 16), m.group(1), m.group(2), base + int(t.group(1), 16) if t else None))
    return funcs


@pytest.fixture(scope="module")
def isa(nb, tmp_path_factory):
    return disassemble(LIB, str(tmp_path_factory.mktemp("isa")))


def _is_result_store(op, args):
    return op.startswith(("global_store", "global_atomic", "flat_store", "flat_atomic", "buffer_store", "buffer_atomic"))


def _drains_vm(op, args):
    # "s_waitcnt vmcnt(0)" alone or with the other counters; a bare "s_waitcnt 0"-style encoding prints all three counters as (0)
    return op == "s_waitcnt" and "vmcnt(0)" in args


def check_drain_before_final_barrier(name, insns):
    """every wave executes an s_waitcnt vmcnt(0) after its last result store and before the workgroup's final barrier"""
    bar = max(k for k, (_, op, _, _) in enumerate(insns) if op == "s_barrier")
    stores = [k for k in range(bar) if _is_result_store(insns[k][1], insns[k][2])]
    assert stores, name + ": no result store in front of the final barrier"
    waits = [k for k in range(stores[-1] + 1, bar) if _drains_vm(insns[k][1], insns[k][2])]
    assert waits, (f"{name}: no s_waitcnt vmcnt(0) between the last result store (0x{insns[stores[-1]][0]:x}) and the final "
                   f"s_barrier (0x{insns[bar][0]:x}): the count-out can overtake this wave's stores")
    wk = waits[-1]
    # ... on the path EVERY wave takes: nothing branches away between the wait and the barrier, nothing jumps in behind the wait
    for k in range(wk, bar):
        assert not insns[k][1].startswith(("s_cbranch", "s_branch", "s_endpgm", "s_setpc")), (name, "branch between the drain and the barrier", insns[k])
    lo, hi = insns[wk][0], insns[bar][0]
    for addr, op, args, tgt in insns:
        assert tgt is None or not (lo < tgt <= hi), (name, f"a branch at 0x{addr:x} enters between the drain and the barrier")
    return bar


def test_library_holds_the_in_place_fused_step_kernels(isa):
    names = [n for n in isa if "step_fused" in n]
    assert len([n for n in names if n.endswith("Lb1EEEvNS_11FusedParamsE")]) == 16, names
    assert len([n for n in names if n.endswith("Lb0EEEvNS_11FusedParamsE")]) == 16, names


def test_in_place_fused_step_drains_its_result_stores_before_the_count_out(isa):
    names = sorted(n for n in isa if "step_fused" in n and n.endswith("Lb1EEEvNS_11FusedParamsE"))
    assert names
    for name in names:
        insns = isa[name]
        bar = check_drain_before_final_barrier(name, insns)
        # behind the barrier: the `finished` count-out (an atomic with return), later the host-mapped word (the last system-scope
        # 64-bit store of the kernel); between the repair path's stores and that word another full drain
        outs = [k for k in range(bar, len(insns)) if insns[k][1].startswith("global_atomic_add") and "sc0" in insns[k][2]]
        assert outs, (name, "no count-out atomic behind the final barrier")
        words = [k for k in range(outs[0], len(insns)) if insns[k][1] == "global_store_dwordx2" and "sc0 sc1" in insns[k][2]]
        assert words, (name, "no host-mapped word store")
        word = words[-1]
        repair = [k for k in words[:-1]]
        if repair:
            assert any(_drains_vm(insns[k][1], insns[k][2]) for k in range(repair[-1] + 1, word)), (name, "repair stores not drained before the host word")


def test_in_place_results_and_the_host_words_are_system_scope_stores(isa):
    """The host is told while the launch is still winding down, and eight XCDs' L2s are not coherent for ordinary stores: every
    result store of the in-place kernels (x, v, a, the spare array, the marks) and the host-mapped word itself must go THROUGH the
    L2 to memory (`sc0 sc1`); a plain `global_store_dwordx4` of a result would sit in one XCD's L2 when the host reads. Same for
    the one-store kernel `host_signal` that stands in for a stream synchronisation above 8192 bodies."""
    for name in sorted(n for n in isa if "step_fused" in n and n.endswith("Lb1EEEvNS_11FusedParamsE")):
        stores = [(op, args) for _, op, args, _ in isa[name] if op.startswith("global_store")]
        assert len(stores) >= 8, (name, stores)
        for op, args in stores:
            if op == "global_store_dword":      # the counters' reset and the running total: agent scope (sc1) or plain, device-side bookkeeping only
                continue
            assert "sc0 sc1" in args, (name, op, args)
        assert not any(op in ("global_store_dwordx4", "global_store_dwordx3") for op, _ in stores), name
    sig = [n for n in isa if "host_signal" in n]
    assert len(sig) == 1
    stores = [(op, args) for _, op, args, _ in isa[sig[0]] if op.startswith("global_store")]
    assert len(stores) == 1 and stores[0][0] == "global_store_dwordx2" and "sc0 sc1" in stores[0][1], stores


def test_the_checker_itself_sees_a_missing_drain():
    """the rule is not vacuous: the round-4 code shape (stores, lgkm-only wait, barrier) fails it; the fixed shape passes"""
    def mk(lines):
        return [(0x100 + 4 * k, op, args, None) for k, (op, args) in enumerate(lines)]
    bad = mk([("global_store_dwordx2", "v[8:9], v[10:11], off sc0 sc1"), ("global_atomic_add", "v0, v1, s[12:13] offset:4"),
              ("s_waitcnt", "lgkmcnt(0)"), ("s_barrier", ""), ("s_waitcnt", "vmcnt(0)"), ("s_endpgm", "")])
    with pytest.raises(AssertionError):
        check_drain_before_final_barrier("bad", bad)
    good = mk([("global_store_dwordx2", "v[8:9], v[10:11], off sc0 sc1"), ("s_waitcnt", "vmcnt(0) expcnt(0) lgkmcnt(0)"), ("s_barrier", ""), ("s_endpgm", "")])
    check_drain_before_final_barrier("good", good)
    skipped = mk([("s_cbranch_execz", "2"), ("global_store_dwordx2", "v[8:9], v[10:11], off sc0 sc1"), ("s_waitcnt", "vmcnt(0)"), ("s_nop", "0"), ("s_barrier", "")])
    skipped[0] = (skipped[0][0], "s_cbranch_execz", "2", skipped[3][0])          # jumps in behind the wait: waves that skipped are fine, but flag the shape
    with pytest.raises(AssertionError):
        check_drain_before_final_barrier("skipped", skipped)


# ---- the second cross-workgroup protocol: block pairs with the sums added in place (nbk::force_sym_ticket) ------------------------------

ACC_STORES = ("global_store_dwordx2", "buffer_store_dwordx4")   # how a kernel may write a partial sum back: agent-scope 64-bit halves, or one 128-bit buffer store
ACC_LOADS = ("global_load_dwordx2", "buffer_load_dwordx4", "buffer_load_dwordx3")   # (x, y, z of a sum are what is added: the compiler may leave w unread)


def _ticket_kernels(isa):
    return sorted(n for n in isa if "force_sym_ticket" in n)


def check_ticket_hand_over(name, insns, waves):
    """Every store that hands a ticket on (global_store_dword sc1 outside the spin loop) comes after: the workgroup's last accumulation
    store (global_store_dwordx2 sc1), an s_waitcnt vmcnt(0) executed by every wave, and — several waves — the barrier. Returns the
    number of hand-over stores found."""
    handed = 0
    for k, (addr, op, args, _) in enumerate(insns):
        if op != "global_store_dword" or "sc1" not in args or "sc0" in args:
            continue
        back = insns[max(0, k - 60):k]
        last_acc = max((q for q, (_, o, a, _) in enumerate(back) if o in ACC_STORES and "sc1" in a), default=None)
        spin = max((q for q, (_, o, _, _) in enumerate(back) if o == "s_memrealtime"), default=None)
        if spin is not None and (last_acc is None or spin > last_acc):
            continue                                             # the abort flag, raised inside the spin loop: not a hand-over
        assert last_acc is not None, (name, f"ticket store at 0x{addr:x} with no accumulation store in front of it")
        tail = back[last_acc + 1:]
        waits = [q for q, (_, o, a, _) in enumerate(tail) if _drains_vm(o, a)]
        assert waits, (name, f"no s_waitcnt vmcnt(0) between the last accumulation store and the ticket store at 0x{addr:x}: the next "
                             "contributor could load sums that are still on their way")
        if waves > 1:
            bars = [q for q, (_, o, _, _) in enumerate(tail) if o == "s_barrier"]
            assert bars and bars[-1] > waits[0], (name, f"the ticket store at 0x{addr:x} is not behind a barrier that follows the drain")
        handed += 1
    return handed


def test_in_place_block_pair_kernels_drain_their_sums_before_handing_the_ticket_on(isa):
    names = _ticket_kernels(isa)
    assert len(names) == 2, names                                 # <SymPacked<10>, 4> and <SymPacked<10>, 1>
    for name in names:
        waves = 4 if "ELi4EEE" in name else 1
        handed = check_ticket_hand_over(name, isa[name], waves)
        assert handed >= 4, (name, handed)                        # I side and J side, general and equal-mass path


def test_in_place_block_pair_kernels_move_their_sums_at_agent_scope(isa):
    """The eight XCDs' L2s are not coherent for ordinary accesses: every sum a ticket kernel writes is an agent-scope store
    (global_store_dwordx2 sc1, through the L2), every sum it reads back an agent-scope load; it writes no slab at all (no plain
    global_store of partial sums), and its spin loop looks at the tickets with agent-scope loads."""
    for name in _ticket_kernels(isa):
        insns = isa[name]
        stores = [(op, args) for _, op, args, _ in insns if op.startswith(("global_store", "buffer_store", "flat_store"))]
        assert stores and all("sc1" in args for _, args in stores), (name, [s for s in stores if "sc1" not in s[1]][:3])
        acc_stores = [1 for op, args in stores if op in ACC_STORES]
        acc_loads = [1 for _, op, args, _ in insns if op in ACC_LOADS and "sc1" in args]
        per_sum = 1 if any(op == "buffer_store_dwordx4" for op, _ in stores) else 2          # one 128-bit access per partial sum, or two 64-bit halves
        want = per_sum * 2 * (10 + 1)                                                         # x (general, equal-mass) x (10 I-side unrolled + the J-side loop)
        assert len(acc_stores) >= want and len(acc_loads) >= want, (name, len(acc_stores), len(acc_loads), want)
        assert not any(op in ACC_LOADS and "sc1" not in args and op.startswith("buffer") for _, op, args, _ in insns), name   # no plain access to the lanes
        polls = [1 for _, op, args, _ in insns if op == "global_load_dword" and "sc1" in args]
        assert len(polls) >= 4, (name, len(polls))
        assert any(op == "s_memrealtime" for _, op, _, _ in insns) and any(op == "s_sleep" for _, op, _, _ in insns), name   # bounded, polite spin


def test_the_ticket_checker_sees_a_missing_drain():
    def mk(lines):
        return [(0x100 + 4 * k, op, args, None) for k, (op, args) in enumerate(lines)]
    good = mk([("global_store_dwordx2", "v[0:1], v[2:3], off sc1"), ("s_waitcnt", "vmcnt(0)"), ("s_barrier", ""), ("global_store_dword", "v0, v1, s[2:3] sc1")])
    assert check_ticket_hand_over("good", good, 4) == 1
    wide = mk([("buffer_store_dwordx4", "v[0:3], v4, s[4:7], 0 offen sc1"), ("s_waitcnt", "vmcnt(0)"), ("s_barrier", ""), ("global_store_dword", "v0, v1, s[2:3] sc1")])
    assert check_ticket_hand_over("wide", wide, 4) == 1
    no_drain = mk([("global_store_dwordx2", "v[0:1], v[2:3], off sc1"), ("s_waitcnt", "lgkmcnt(0)"), ("s_barrier", ""), ("global_store_dword", "v0, v1, s[2:3] sc1")])
    with pytest.raises(AssertionError):
        check_ticket_hand_over("no_drain", no_drain, 4)
    no_barrier = mk([("global_store_dwordx2", "v[0:1], v[2:3], off sc1"), ("s_waitcnt", "vmcnt(0)"), ("global_store_dword", "v0, v1, s[2:3] sc1")])
    with pytest.raises(AssertionError):
        check_ticket_hand_over("no_barrier", no_barrier, 4)
    assert check_ticket_hand_over("one wave", no_barrier, 1) == 1
    abort = mk([("s_memrealtime", "s[4:5]"), ("global_store_dword", "v0, v1, s[2:3] sc1"), ("global_store_dword", "v0, v1, s[6:7] sc0 sc1")])
    assert check_ticket_hand_over("abort", abort, 4) == 0
