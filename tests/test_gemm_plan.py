"""CPU: the work plan of tiles 26 / 27 (rga3_gemm_ragged_plan, host only) against a restatement of gemm_nt_sk_kernel's work-list decode: over all workgroups every
(tile, K-iteration) is computed exactly once; a tile is cut into at most three slices, the owner (last slice) has the highest workgroup number and its contributors
are the nearest lower-numbered workgroups with a run (empty runs skipped); each workgroup writes at most one slab; ragged tiles have workgroups of their own; and the
predicted makespan is not worse than tile 22's padded plan.
No reference counterpart (the reference's GEMMs are vendor BLAS calls: HF modeling_qwen2_5_vl.py:211-321 via reference model/qwen_2_5_vl_sam2.py:182-200)."""
import ctypes as C

import pytest

from rga3.hip import lib

SHAPES = [(2112, 37888, 3584), (2112, 3584, 18944), (2112, 4608, 3584), (2112, 3584, 3584), (2112, 18944, 3584), (2112, 3584, 37888), (2112, 152064, 3584),
          (4160, 37888, 3584), (4160, 3584, 18944), (4160, 4608, 3584), (320, 512, 128), (2050, 1024, 3584), (1088, 16384, 1024), (65832, 1280, 1280),
          (2112, 3840, 1280), (320, 256, 64), (576, 768, 6400)]


def plan(M, N, K, cus, split=1):
    so = lib.load()
    pl = (C.c_int * 8)()
    st = (C.c_uint * (cus + 1))()
    rc = so.rga3_gemm_ragged_plan(M, N, K, cus, split, pl, st)
    return rc, list(pl), list(st)


def work_list(w, P, nk, ntn, t_dp, sk_tiles, G, start):
    """gemm_nt_sk_kernel's decode: [(tile, kb, ke, kind)] of workgroup w in execution order (kind 0 whole, 1 non-owner slice, 2 owner slice)."""
    na = ow = None
    tw0 = n_tw = 0
    if sk_tiles > 0:
        x0, x1 = start[w], start[w + 1]
        if x0 < x1:
            ta, tz = x0 // nk, (x1 - 1) // nk
            sa, ez = ta * nk, (tz + 1) * nk
            first_whole, end_whole = ta, tz + 1
            if x0 > sa:
                first_whole = ta + 1
                if x1 >= sa + nk:
                    ow = (t_dp + ta, x0 - sa, nk, 2)
                else:
                    na = (t_dp + ta, x0 - sa, x1 - sa, 1)
            if x1 < ez and (tz > ta or x0 == sa):
                end_whole = tz
                na = (t_dp + tz, 0, x1 - tz * nk, 1)
            tw0, n_tw = t_dp + first_whole, max(end_whole - first_whole, 0)
    if G > 0:
        below = (w * G) // P
        if ((w + 1) * G) // P > below:            # a workgroup of ragged tiles only: columns below, below + G, ...
            dp = list(range(below, ntn, G))
        else:
            rk, Pf = w - below, P - G
            dp = list(range(ntn + rk, t_dp, Pf))
    else:
        dp = list(range(w, t_dp, P))
    items = [na] if na else []
    items += [(t, 0, nk, 0) for t in dp]
    items += [(tw0 + i, 0, nk, 0) for i in range(n_tw)]
    if ow:
        items.append(ow)
    return items


@pytest.mark.parametrize("split", [1, 0])
@pytest.mark.parametrize("cus", [256, 304, 64])
@pytest.mark.parametrize("M,N,K", SHAPES)
def test_ragged_plan_covers_every_iteration_once(M, N, K, cus, split):
    rc, (t_dp, sk_tiles, ragged, G, _, P, P_sk, gm), start = plan(M, N, K, cus, split)
    assert rc in (0, 1)
    if rc == 1:
        return
    ntm, ntn, nk = -(-M // 256), -(-N // 256), K // 64
    T = ntm * ntn
    assert ragged == 1 and t_dp + sk_tiles == T and P <= cus and G > 0 and T >= cus
    assert start[0] == 0 and start[cus] == sk_tiles * nk and all(a <= b for a, b in zip(start, start[1:]))
    if not split:
        assert sk_tiles == 0
    seen = {}
    span = 0.0
    for w in range(P):
        items = work_list(w, P, nk, ntn, t_dp, sk_tiles, G, start)
        assert sum(1 for it in items if it[3] == 1) <= 1            # one slab per workgroup
        assert [it[3] for it in items] == sorted((it[3] for it in items), key=lambda k: (0 if k == 1 else 2 if k == 2 else 1))   # non-owner first, owner last
        cost = 0.0
        for tile, kb, ke, kind in items:
            assert 0 <= tile < T and 0 <= kb < ke <= nk
            seen.setdefault(tile, []).append((kb, ke, kind, w))
            cost += (ke - kb) * (0.625 if tile < ntn else 1.0)
        span = max(span, cost)
    assert set(seen) == set(range(T))
    for tile, sl in seen.items():
        sl.sort()
        assert sl[0][0] == 0 and sl[-1][1] == nk and all(a[1] == b[0] for a, b in zip(sl, sl[1:])), (tile, sl)      # a partition of [0, nk)
        assert len(sl) <= 3, (tile, sl)
        if len(sl) == 1:
            assert sl[0][2] == 0
        else:
            assert [x[2] for x in sl] == [1] * (len(sl) - 1) + [2] and [x[3] for x in sl] == sorted(x[3] for x in sl), (tile, sl)
            # the owner finds its contributors as the nearest lower-numbered workgroups with a run
            owner = sl[-1][3]
            c = owner - 1
            found = []
            while len(found) < len(sl) - 1:
                if start[c] < start[c + 1]:
                    found.append(c)
                c -= 1
            assert sorted(found) == [x[3] for x in sl[:-1]], (tile, sl, found)
    # makespan in full K-iterations against tile 22's (every tile at full cost: whole rounds + equal runs over P_sk = min(2 rem, rem nk / 8, cus) workgroups)
    rem = T % cus
    p_sk = max(min(2 * rem, rem * nk // 8, cus), rem)
    span22 = (T // cus) * nk + (-(-rem * nk // p_sk) if rem else 0)
    if split:
        assert span <= span22 + 1, (span, span22)


def test_model_shapes_gain():
    """gate | up and the LM head at M = 2112 (a ragged tile row in nine) on workgroups of their own for the ragged tiles: five rounds for gate | up (tile 22: five and a
    half), under 20.7 for the LM head (21 padded); the ragged workgroups walk the tile columns in step with the full-tile ones."""
    for (M, N, K), want in (((2112, 37888, 3584), 5.0), ((2112, 152064, 3584), 20.7)):
        rc, (t_dp, sk_tiles, ragged, G, _, P, P_sk, gm), start = plan(M, N, K, 256, 0)
        assert rc == 0 and G > 0
        ntm, ntn, nk = -(-M // 256), -(-N // 256), K // 64
        lists = [work_list(w, P, nk, ntn, t_dp, sk_tiles, G, start) for w in range(P)]
        span = max(sum((ke - kb) * (0.625 if t < ntn else 1.0) for t, kb, ke, _ in items) for items in lists)
        assert span <= want * nk + 1, (M, N, K, span / nk)
        rag = [items for items in lists if items and items[0][0] < ntn]
        assert len(rag) == G and all(t < ntn for items in rag for t, *_ in items)
        # the k-th ragged tile of a workgroup is column rank + k G; the full tiles of that column are reached (P - G) / (ntm - 1) columns per round
        cols_per_round = (P - G) / (ntm - 1)
        for items in rag:
            for k, (t, *_rest) in enumerate(items):
                assert abs(k * 0.625 - t / cols_per_round) <= 1.0, (k, t)


def test_non_ragged_products_have_no_plan():
    assert plan(2048, 4096, 1024, 256)[0] == 1      # no padded tile row
    assert plan(2113, 4096, 1024, 256)[0] == 1      # 65 rows in the last tile row
    assert plan(64, 4096, 1024, 256)[0] == 1        # a single tile row
    assert plan(2112, 4096, 1000, 256)[0] < 0       # K not a multiple of 64
