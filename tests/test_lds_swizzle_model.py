"""CPU: the LDS bank model of tools/probes/lds_swizzle_sim.py applied to the slab swizzle used by the vocoder and
predictor kernels (SlabSwizzle, csrc/fs2_common.h): an MFMA fragment read (lane (fr, fg) -> row r0 + fr, slot
kc*4 + fg, ds_read_b128) must be conflict-free for EVERY start row, and the map and its inverse (used by the
LDS-DMA fills, which fetch the logical slot that belongs at a physical position) must agree."""
import pytest

GROUPS = [list(range(0, 4)) + list(range(12, 16)) + list(range(20, 28)),
          list(range(4, 12)) + list(range(16, 20)) + list(range(28, 32)),
          list(range(32, 36)) + list(range(44, 48)) + list(range(52, 60)),
          list(range(36, 44)) + list(range(48, 52)) + list(range(60, 64))]  # MI355X_MICROARCH.md, LDS table


def slot(L, row, ns):  # SlabSwizzle::slot
    nb = min(ns, 16); half = nb // 2; sh = {16: 0, 8: 1, 4: 2}[nb]
    return (L & ~(nb - 1)) | ((L & 1) * half) | ((((L & (nb - 1)) >> 1) ^ (row >> sh)) & (half - 1))


def logical(ps, row, ns):  # SlabSwizzle::logical
    nb = min(ns, 16); half = nb // 2; sh = {16: 0, 8: 1, 4: 2}[nb]
    return (ps & ~(nb - 1)) | ((((ps & (half - 1)) ^ ((row >> sh) & (half - 1))) << 1) | (1 if ps & half else 0))


def read_cycles(addrs):
    c = 0
    for g in GROUPS:
        quads = {}
        for l in g:
            quads.setdefault((addrs[l] // 16) % 16, set()).add(addrs[l])
        c += max(len(v) for v in quads.values())
    return c


@pytest.mark.parametrize("ns", [4, 8, 16, 32, 64])
def test_fragment_reads_are_conflict_free_for_every_start_row(ns):
    rowb = ns * 16
    for r0 in range(64):
        for kc in range(ns // 4):
            addrs = [(r0 + (l & 15)) * rowb + slot(kc * 4 + (l >> 4), r0 + (l & 15), ns) * 16 for l in range(64)]
            assert read_cycles(addrs) == 4, (ns, r0, kc)  # four lane groups, one LDS cycle each


@pytest.mark.parametrize("ns", [4, 8, 16, 32, 64])
def test_swizzle_is_a_bijection_and_logical_is_its_inverse(ns):
    for row in range(128):
        phys = [slot(L, row, ns) for L in range(ns)]
        assert sorted(phys) == list(range(ns))
        for L in range(ns):
            assert logical(slot(L, row, ns), row, ns) == L
        # fragments 16 rows apart share the swizzle (the kernels add mi*16*rowb to one base address)
        assert all(slot(L, row + 16, ns) == slot(L, row, ns) for L in range(ns))
