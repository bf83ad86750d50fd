#!/usr/bin/env python3
"""Step-level model of gemm_bf16_256x.hip's schedule (the persistent GEMM whose two wave groups alternate between MFMAs
and requests + epilogue): the same programs, frontier cursor and ring-slot rules as the kernel, executed barrier by
barrier for one workgroup, with every operand request and every LDS use checked:

  * both wave groups execute the same number of barriers (a mismatch would hang the workgroup),
  * in every step at most one group issues MFMAs, and never a group that is in its epilogue,
  * every stage is requested exactly once, two steps before it is read, by waves that are not in MFMAs, into the ring
    slot (stage mod 3) that nobody reads in that step, and the requesting waves execute a covering wait before the
    barrier in front of the step that reads it,
  * a computing group finds ITS half tile's rows and the tile's weight panel at K slice 0, 1, .. nk-1 in that order
    (the accumulation order of gemm_bf16_nt_256p_kernel: results are bit-identical).

Run: python scripts/pingpong_model.py      Used by tests/test_pingpong_model.py."""
from __future__ import annotations


def tile_list(wg, grid, tiles_m, tiles_n, group_m=4):
    """tile list of workgroup `wg` as tile_origin() of the persistent kernels lays it out: (m0, n0) in rows / columns"""
    ntiles = tiles_m * tiles_n
    out = []
    for tile in range(wg, ntiles, grid):
        q8, r8, xcd, loc = ntiles >> 3, ntiles & 7, tile & 7, tile >> 3
        t = (xcd * (q8 + 1) if xcd < r8 else r8 * (q8 + 1) + (xcd - r8) * q8) + loc
        gs = group_m * tiles_n
        first_m = (t // gs) * group_m
        gm = min(tiles_m - first_m, group_m)
        out.append(((first_m + (t % gs) % gm) * 256, ((t % gs) // gm) * 256))
    return out


def programs(ntw, nk, E):
    """per group, what it does in each of its steps: ('idle',), ('mma', tile, kt), ('epi', tile, chunk)"""
    g0, g1 = [], [("idle",)] * nk
    for ti in range(ntw):
        g0 += [("mma", ti, kt) for kt in range(nk)] + [("epi", ti, c) for c in range(E)] + [("idle",)] * (nk - E)
        off = E if ti + 1 == ntw else nk
        g1 += [("mma", ti, kt) for kt in range(nk)] + [("epi", ti, c) for c in range(E)] + [("idle",)] * (off - E)
    g0 += [("idle",)] * E
    return g0, g1


def simulate(tiles, K, E):
    nk, ntw = K // 64, len(tiles)
    assert nk >= 8 and nk % 2 == 0 and E <= nk
    progs = programs(ntw, nk, E)
    assert len(progs[0]) == len(progs[1]), "barrier counts differ: the workgroup would hang"
    G = len(progs[0])
    # frontier cursor exactly as the kernel keeps it
    f = dict(k=0, grp=0, tile=0, slot=0)

    def frontier_step():
        f["slot"] = (f["slot"] + 1) % 3
        f["k"] += 1
        if f["k"] == nk:
            f["k"] = 0
            f["grp"] ^= 1
            if f["grp"] == 0:
                f["tile"] += 1

    slots = [None, None, None]           # (stage, m_rows, n0, k, requested_in_step, covered_by_barrier)

    def request(step, who):
        if f["tile"] >= ntw:
            return False
        m0, n0 = tiles[f["tile"]]
        stage = step + 2
        prev = slots[f["slot"]]
        assert prev is None or prev[0] == stage - 3, (step, "slot not free", prev)
        assert f["slot"] == stage % 3
        # requested during `step`; the requester's next end-of-step wait (step + 1) covers it; vmcnt(0) at the end of a
        # compute step if the requester has turned to MFMAs by then
        slots[f["slot"]] = (stage, m0 + 128 * f["grp"], n0, f["k"], step, step + 1, who)
        return True

    # prologue: stages 0 and 1 by all waves, waited
    for st in (0, 1):
        m0, n0 = tiles[0]
        slots[st] = (st, m0, n0, st, -1, -1, "all")
        frontier_step()
    assert f["slot"] == 2 and f["k"] == 2
    seen = {}
    stats = dict(steps=G, mma=0, none=0)
    for g in range(G):
        act = (progs[0][g], progs[1][g])
        computing = [X for X in (0, 1) if act[X][0] == "mma"]
        assert len(computing) <= 1, (g, "both groups in MFMAs")
        stats["mma" if computing else "none"] += 1
        requested = 0
        for X in (0, 1):
            if act[X][0] != "mma":
                requested += request(g, X) if requested == 0 else 0
        # (both non-computing groups call request() in the kernel; only when no stage is left, so at most one issues)
        if not computing and f["tile"] < ntw:
            raise AssertionError((g, "a step without MFMAs while stages remain"))
        for X in computing:
            _, ti, kt = act[X]
            m0, n0 = tiles[ti]
            s = slots[g % 3]
            assert s is not None and s[0] == g, (g, "stage not in its slot", s)
            assert s[1] == m0 + 128 * X and s[2] == n0 and s[3] == kt, (g, X, s, (m0, n0, kt))
            assert s[5] <= g - 1, (g, "stage read before its requester's covering wait")
            seen.setdefault((X, ti), []).append(s[3])
        frontier_step()
    for X in (0, 1):
        for ti in range(ntw):
            assert seen.get((X, ti)) == list(range(nk)), (X, ti, seen.get((X, ti)))
    return stats


def epilogue_steps(epi, has_pre=True):
    """E of gemm_bf16_nt_256x_kernel: 8 for the fp32 epilogues and the two-output activation pair, else 4"""
    return 8 if epi in (1, 4) or (epi == 2 and has_pre) else 4


if __name__ == "__main__":
    for (M, N, K, epi) in [(32768, 3072, 1024, 0), (32768, 1024, 1024, 1), (32768, 4096, 1024, 2), (32768, 1024, 4096, 1),
                           (32768, 4096, 1024, 3), (32768, 1024, 3072, 0), (16384, 1024, 512, 4), (2560, 7680, 1024, 0),
                           (512, 256, 512, 0)]:
        tm, tn = M // 256, N // 256
        grid = min(tm * tn, 256)
        E = epilogue_steps(epi)
        cover = []
        for wg in range(grid):
            tl = tile_list(wg, grid, tm, tn)
            st = simulate(tl, K, E)
            cover += tl
        assert sorted(cover) == sorted((m * 256, n * 256) for m in range(tm) for n in range(tn))
        print(f"M={M} N={N} K={K} epi={epi} E={E}: grid {grid}, {st['steps']} steps in the last workgroup "
              f"({st['mma']} with MFMAs, {st['none']} without): OK")
