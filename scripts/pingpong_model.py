#!/usr/bin/env python3
"""Step-level model of gemm_bf16_256x.hip's schedule (the phase-shifted persistent GEMM): the same cursor / role /
ring-slot rules as the kernel, executed barrier by barrier for one workgroup, with every operand request and every
LDS use checked:

  * both wave groups execute the same number of barriers (a mismatch would hang the workgroup),
  * a stage's B panel slice and each computing group's A half are requested exactly once, into a slot that nobody reads
    or stages into at that moment, and have "landed" (their requester waited) by the barrier before they are read,
  * a computing group always finds ITS tile's rows / panel at the K slice of the current step, and every tile of every
    group sees each of the nk K slices exactly once (a rotation of 0 .. nk-1),
  * the epilogue staging area (the group's half of the current stage's A slot) is never a request target of that step.

Run: python scripts/pingpong_model.py            (the encoder's shapes at B = 128 + a few odd ones)
Used by tests/test_pingpong_model.py."""
from __future__ import annotations

import itertools


def tile_lists(tiles_m, tiles_n):
    """tile list per workgroup as gemm_bf16_nt_256x() lays it out; None when the shape does not qualify"""
    nb = 4 if tiles_n % 4 == 0 else 2 if tiles_n % 2 == 0 else 1
    mb = 32 // nb
    if tiles_m % mb:
        return None
    mblocks = tiles_m // mb
    blocks = mblocks * (tiles_n // nb)
    if blocks % 8:
        return None
    bpx = blocks // 8
    out = {}
    for wg in range(256):
        xl, x = wg >> 3, wg & 7
        li, lj = xl % mb, xl // mb
        out[wg] = [((((x * bpx + t) % mblocks) * mb + li) * 256, (((x * bpx + t) // mblocks) * nb + lj) * 256) for t in range(bpx)]
    return out


class Cur:
    def __init__(self, tiles, ph, P, stall_len):
        self.tiles, self.P, self.stall_len = tiles, P, stall_len
        self.tile, self.ph = 0, ph
        self.load()

    def load(self):
        T = len(self.tiles)
        if self.tile < T:
            self.m0, self.n0 = self.tiles[self.tile]
            nn = self.tiles[self.tile + 1][1] if self.tile + 1 < T else self.n0
            self.per = self.P + (self.stall_len if nn != self.n0 else 0)
        else:
            self.per = 1 << 30

    def computes(self, nk):
        return self.tile < len(self.tiles) and 0 <= self.ph < nk

    def step(self):
        self.ph += 1
        if self.ph == self.per:
            self.tile += 1
            self.ph = 0
            self.load()


def own_program(X, tiles, nk, E, H, stall_len):
    """what group X does in each of its steps, in order: ('idle',), ('mma', tile, kt), ('epi', tile, chunk)"""
    prog = [("idle",)] * (H if X else 0)
    for ti, (m0, n0) in enumerate(tiles):
        prog += [("mma", ti, kt) for kt in range(nk)]
        prog += [("epi", ti, c) for c in range(E)]
        stall = stall_len if ti + 1 < len(tiles) and tiles[ti + 1][1] != n0 else 0
        prog += [("idle",)] * stall
    prog += [("idle",)] * (0 if X else H)
    return prog


def simulate(tiles, K, E):
    nk = K // 64
    assert nk % 2 == 0 and E % 2 == 0 and E <= nk and nk >= 8
    P, H = nk + E, (nk + E) // 2
    stall_len = H - E
    progs = [own_program(X, tiles, nk, E, H, stall_len) for X in (0, 1)]
    assert len(progs[0]) == len(progs[1]), "barrier counts differ: the workgroup would hang"
    G = len(progs[0])
    cur = [Cur(tiles, 0, P, stall_len), Cur(tiles, -H, P, stall_len)]
    c0 = [c.computes(nk) for c in cur]
    for c in cur:
        c.step()
    c1 = [c.computes(nk) for c in cur]
    n1 = [c.n0 for c in cur]
    for c in cur:
        c.step()
    c2 = [c.computes(nk) for c in cur]
    n2 = [c.n0 for c in cur]
    m2 = [c.m0 for c in cur]
    # LDS contents: a_slot[s][Z] = (stage, m0, kslice, landed_by_barrier) ; b_slot[s] = (stage, n0, kslice, landed_by_barrier)
    a_slot = [[None, None] for _ in range(3)]
    b_slot = [None, None]
    # prologue (all waves, waited before the first barrier): B(0), group 0's half of A(0), A(1)
    b_slot[0] = (0, tiles[0][1], 0, -1)
    a_slot[0][0] = (0, tiles[0][0], 0, -1)
    a_slot[1][0] = (1, tiles[0][0], 1 % nk, -1)
    kb1, kb2 = 1 % nk, 2 % nk
    seen = {}                                  # (X, tile) -> list of k slices accumulated
    stats = dict(steps=G, cc=0, alone=0, none=0)
    for g in range(G):
        act = [progs[0][g], progs[1][g]]
        # flags and own programs must tell the same story
        for Z in (0, 1):
            assert c0[Z] == (act[Z][0] == "mma"), (g, Z, c0, act)
        cc = c0[0] and c0[1]
        stats["cc" if cc else "alone" if (c0[0] or c0[1]) else "none"] += 1
        sa0, sa2, sb0, sb1 = g % 3, (g + 2) % 3, g % 2, (g + 1) % 2
        staging = [(sa0, Z) for Z in (0, 1) if act[Z][0] == "epi"]
        reads_a = [(sa0, Z) for Z in (0, 1) if c0[Z]]
        # ---- requests of this step (duties_after_barrier / duties_end_of_step): who, what
        requester = 1 if cc else (1 if c0[0] else 0)          # B (and A in a one-group step)
        if c1[0] or c1[1]:
            nb1 = n1[0] if c1[0] else n1[1]
            if c1[0] and c1[1]:
                assert n1[0] == n1[1], (g, "two panels wanted in one stage")
            prev = b_slot[sb1]
            assert prev is None or prev[0] <= g - 1, (g, "B slot still to be read", prev)
            assert not (c0[0] or c0[1]) or sb1 != sb0
            b_slot[sb1] = (g + 1, nb1, kb1, g)               # landed by barrier g (the requester waits in this step)
        for Z in (0, 1):
            if c2[Z]:
                assert (sa2, Z) not in staging and (sa2, Z) not in reads_a, (g, "A request into a half in use", Z)
                prev = a_slot[sa2][Z]
                assert prev is None or prev[0] <= g - 1, (g, "A half still to be read", prev)
                # both-compute step: group 0 requests at the END of the step, 8 pieces may stay in flight -> landed by
                # barrier g+1 (its next end-of-step wait); one-group step: requested right after the barrier, waited in-step
                a_slot[sa2][Z] = (g + 2, m2[Z], kb2, g + 1 if cc else g)
        # ---- what the computing groups read in this step
        for Z in (0, 1):
            if not c0[Z]:
                continue
            _, ti, kt = act[Z]
            m0, n0 = tiles[ti]
            a = a_slot[sa0][Z]
            b = b_slot[sb0]
            assert a is not None and a[0] == g and a[1] == m0 and a[2] == g % nk and a[3] <= g - 1, (g, Z, "A", a, m0)
            assert b is not None and b[0] == g and b[1] == n0 and b[2] == g % nk and b[3] <= g - 1, (g, Z, "B", b, n0)
            seen.setdefault((Z, ti), []).append(g % nk)
        # staging halves must not hold a stage somebody requested for this step
        for (s, Z) in staging:
            a = a_slot[s][Z]
            assert a is None or a[0] < g, (g, "staging over a live A half", a)
        # ---- advance
        c0, c1 = c1, c2
        n1 = n2
        for c in cur:
            c.step()
        c2 = [c.computes(nk) for c in cur]
        n2 = [c.n0 for c in cur]
        m2 = [c.m0 for c in cur]
        kb1, kb2 = kb2, (kb2 + 1) % nk
    for X in (0, 1):
        for ti in range(len(tiles)):
            ks = seen.get((X, ti))
            assert ks is not None and sorted(ks) == list(range(nk)), (X, ti, ks)
            assert all((b - a) % nk == 1 for a, b in zip(ks, ks[1:])), "K slices of a tile must be consecutive (a rotation)"
    return stats


def epilogue_steps(epi, has_pre=True):
    """E of gemm_bf16_nt_256x_kernel: 8 for the fp32 epilogues and the two-output activation pair, else 4"""
    return 8 if epi in (1, 4) or (epi == 2 and has_pre) else 4


if __name__ == "__main__":
    for (M, N, K, epi) in [(32768, 3072, 1024, 0), (32768, 1024, 1024, 1), (32768, 4096, 1024, 2), (32768, 1024, 4096, 1),
                           (32768, 4096, 1024, 3), (32768, 1024, 3072, 0), (32768, 1024, 4096, 0), (16384, 3072, 1024, 0),
                           (8192, 1024, 512, 4), (4096, 2048, 1024, 0)]:
        tl = tile_lists(M // 256, N // 256)
        if tl is None:
            print(f"M={M} N={N} K={K}: shape does not qualify")
            continue
        E = epilogue_steps(epi)
        agg = None
        for wg, tiles in tl.items():
            st = simulate(tiles, K, E)
            agg = st if agg is None else agg
        # every tile of the problem exactly once
        allt = sorted(itertools.chain.from_iterable(tl.values()))
        assert allt == sorted((m * 256, n * 256) for m in range(M // 256) for n in range(N // 256))
        switches = sum(1 for t in tl.values() for a, b in zip(t, t[1:]) if a[1] != b[1])
        print(f"M={M} N={N} K={K} epi={epi} E={E}: {len(tl[0])} tiles per workgroup, {agg['steps']} steps "
              f"(both groups in MFMAs {agg['cc']}, one {agg['alone']}, none {agg['none']}), panel switches {switches}: OK")
