"""Properties of the generated gfx950 code that the dm | dh kernel's two-chunks-ahead loads depend on
(csrc/gru_bwd128_f16.hip, gru_bwd_dx_deep_f16_kernel).  The loads of its K loop are inline assembly with ONE written-out
wait per chunk; two things have silently broken that before and cost 15 % each time:
  * a wait of the compiler's own inside the loop (for a load of the ragged-tile epilogue that was not consumed on every
    path), which also waits for the loop's in-flight requests;
  * register copies of an in-flight row set in front of the written-out wait.
hipcc cross-compiles without a GPU; the test reads the assembly."""
import os
import re
import subprocess

import pytest

from conftest import REPO

SRC = os.path.join(REPO, "mpnn_amd", "csrc", "gru_bwd128_f16.hip")


@pytest.fixture(scope="module")
def asm(tmp_path_factory):
    out = str(tmp_path_factory.mktemp("isa") / "dx.s")
    r = subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "-fno-fast-math", "-S",
                        "--cuda-device-only", "-I" + os.path.dirname(SRC), "-o", out, SRC], capture_output=True, text=True,
                       timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    return open(out).read()


def _kernel(asm, mangled_args):
    m = re.search(r"^(_ZN4mpnn26gru_bwd_dx_deep_f16_kernel" + mangled_args + r"[^:\n]*):", asm, re.M)
    assert m, mangled_args
    body = asm[m.end():asm.index(".Lfunc_end", m.end())]
    return [ln.strip() for ln in body.split("\n")]


@pytest.mark.parametrize("inst", ["ILi128ELb0EE", "ILi128ELb1EE", "ILi256ELb0EE", "ILi256ELb1EE"])
def test_dx_kernel_k_loop_has_only_the_written_out_wait(asm, inst):
    body = _kernel(asm, inst)
    bars = [i for i, ln in enumerate(body) if ln == "s_barrier"]
    assert len(bars) >= 3                                           # three unrolled chunk bodies
    rows = set()
    pat = re.compile(r"global_load_dwordx4 v\[(\d+):(\d+)\], (v\[\d+:\d+\]), off(?: offset:(\d+))?$")
    for i, ln in enumerate(body):                                   # destinations of the row-set loads: the inline-assembly
        m = pat.match(ln)                                           # groups of four at offsets 0 / 1024 / 2048 / 3072
        if m and m.group(4) is None and i + 3 < len(body):
            grp = [pat.match(body[i + k]) for k in range(4)]
            if all(grp) and [g.group(4) for g in grp] == [None, "1024", "2048", "3072"] and len({g.group(3) for g in grp}) == 1:
                for g in grp:
                    rows |= set(range(int(g.group(1)), int(g.group(2)) + 1))
    assert len(rows) == 48                                          # three sets of sixteen registers
    for a in bars[:3]:
        b = next(i for i in range(a, len(body)) if body[i] in ("s_waitcnt vmcnt(8)", "s_waitcnt vmcnt(6)"))
        region = body[a:b]
        assert not [ln for ln in region if ln.startswith("s_waitcnt vmcnt")], "compiler wait inside the K loop"
        assert sum(ln.startswith("v_mfma") for ln in region) >= 6   # (the straight-line part; the other chunk kinds branch out)
        for ln in region:                                           # no copy / spill of an in-flight row set
            if ln.startswith(("v_mov", "scratch_", "buffer_store")):
                regs = set()
                for lo, hi, one in re.findall(r"v\[(\d+):(\d+)\]|\bv(\d+)\b", ln):
                    regs |= set(range(int(lo), int(hi) + 1)) if lo else {int(one)}
                assert not (regs & rows), ln


@pytest.mark.parametrize("inst", ["ILi128ELb0EE", "ILi128ELb1EE", "ILi256ELb0EE", "ILi256ELb1EE"])
def test_dx_kernel_nothing_touches_a_row_set_in_flight(asm, inst):
    """The compiler takes the inline-assembly loads' outputs as valid at once; what keeps that true is that NO instruction --
    not only a copy -- names a row-set register between the set's four loads and the wait it lands behind.  That wait is
    DERIVED, not a literal: the first s_waitcnt vmcnt(N) with N no larger than the number of requests issued after the set
    (a chunk's own written-out wait leaves the latest eight / six requests, its own among them, in flight; the next chunk's
    lands it).  Checked in text order for every set whose landing wait comes before the loop's way back."""
    body = _kernel(asm, inst)
    pat = re.compile(r"global_load_dwordx4 v\[(\d+):(\d+)\], (v\[\d+:\d+\]), off(?: offset:(\d+))?$")
    vmem = re.compile(r"(global_|buffer_|scratch_)(load|store|atomic)")
    wait = re.compile(r"s_waitcnt vmcnt\((\d+)\)")
    checked = 0
    for i, ln in enumerate(body):
        m = pat.match(ln)
        if not (m and m.group(4) is None and i + 3 < len(body)):
            continue
        grp = [pat.match(body[i + k]) for k in range(4)]
        if not (all(grp) and [g.group(4) for g in grp] == [None, "1024", "2048", "3072"] and len({g.group(3) for g in grp}) == 1):
            continue
        regs = set()
        for g in grp:
            regs |= set(range(int(g.group(1)), int(g.group(2)) + 1))
        # where the set has landed: the first wait that leaves no more requests in flight than were issued after the set
        # (requests counted in text order; a copy loop's body counts once, which can only delay the landing point)
        younger, landed = 0, None
        for k in range(i + 4, len(body)):
            if body[k].startswith("s_cbranch") and re.search(r"\.LBB\d+_\d+$", body[k]) and any(
                    b.startswith(body[k].split()[-1] + ":") for b in body[:i]):
                break                                               # the way back to an earlier label: the next trip takes over
            w = wait.match(body[k])
            if w and int(w.group(1)) <= younger:
                landed = k
                break
            if vmem.match(body[k]):
                younger += 1
        if landed is None:
            continue
        for ln2 in body[i + 4:landed]:
            if not ln2 or ln2.startswith((";", ".", "s_")):
                continue
            named = set()
            for lo, hi, one in re.findall(r"v\[(\d+):(\d+)\]|\bv(\d+)\b", ln2):
                named |= set(range(int(lo), int(hi) + 1)) if lo else {int(one)}
            assert not (named & regs), (ln, ln2)
        checked += 1
    assert checked >= 2
