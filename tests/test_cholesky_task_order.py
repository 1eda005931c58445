"""The task list of the one-launch Cholesky factorisation (csrc/cholesky.hip: k_cholesky_tasks) must be a topological order of
its own dependency graph: a workgroup may only wait for the chain workgroup (resident from the first cycle) and for tasks
dispatched BEFORE it - workgroups are dispatched in list order, so then the lowest unfinished task always has its inputs
complete and the grid cannot deadlock however few workgroups fit the chip.  This replays the list on the host (no GPU) with the
waits the device code performs (PrepTask / SolveTask / the update branch of k_cholesky_tasks) and checks that every counter
value a task waits for has been produced by an earlier task.  Replaces the linear solve inside ceres::Solve
(reference src/optim/bundle_adjustment.cc:273-306); there is no reference counterpart of the schedule itself."""
import ctypes as C

import numpy as np
import pytest

from privacy_preserving_sfm_amd import _capi

PREP_X, PREP_D, SOLVE, UPDATE, PAIR_PREP = 1, 2, 3, 4, 5


def _task_list(T):
    L = _capi.lib()
    n = C.c_int64(0)
    assert L.pp_cholesky_task_list(T, None, 0, C.byref(n)) == 0
    buf = np.zeros(4 * n.value, dtype=np.int32)
    assert L.pp_cholesky_task_list(T, buf.ctypes.data_as(C.POINTER(C.c_int32)), n.value, C.byref(n)) == 0
    return buf.reshape(-1, 4)


def _own(k, r, c):      # the three tiles the chain / the prep tasks of step k update themselves
    return (r == k + 1 and c == k + 1) or (r == k + 2 and c in (k + 1, k + 2))


def _valid(T, k, r, c):
    return r < T and c < T and r >= c and c >= k + 1 and not _own(k, r, c)


@pytest.mark.parametrize("T", [4, 5, 6, 7, 8, 13, 24, 47, 48, 63, 64, 94, 128])
def test_task_list_is_a_topological_order_and_complete(T):
    tasks = _task_list(T)
    sol = np.zeros(T + 4, dtype=np.int64)          # sol[i]: solved columns of block row i
    ver = {}                                        # ver[(I, J)]: panels applied to super-tile (I, J)
    sub = {}
    seen = set()
    solved = set()                                  # (row, column) tiles some task of the list solves
    applied = {}                                    # (r, c) -> panels applied by update tasks, in order

    def need_ver(I, J, v, what):
        assert ver.get((I, J), 0) >= v, "%s waits for ver[%d][%d] >= %d before any earlier task produced it" % (what, I, J, v)

    def need_sol(i, v, what):
        assert sol[i] >= v, "%s waits for sol[%d] >= %d before any earlier task produced it" % (what, i, v)

    for typ, k, a, b in tasks:
        what = "task (type %d, k %d, a %d, b 0x%x)" % (typ, k, a, b)
        key = (int(typ), int(k), int(a), int(b))
        assert key not in seen, what + " listed twice"
        seen.add(key)
        if typ in (PREP_X, PREP_D):
            assert k + 2 < T
            if k > 0:
                need_ver((k + 2) >> 1, k >> 1, k - 1, what)
                need_ver((k + 2) >> 1, ((k + 1) >> 1) if typ == PREP_X else ((k + 2) >> 1), k - 1, what)
                need_sol(k + 2, k, what)
                if typ == PREP_X:
                    need_sol(k + 1, k, what)
            if typ == PREP_X:                       # publishes: tile (k+2,k) solved, the chain's tile (k+1,k) copied to L
                assert sol[k + 2] == k and sol[k + 1] == k
                sol[k + 2] = k + 1; sol[k + 1] = k + 1
                solved.add((k + 2, k)); solved.add((k + 1, k))
        elif typ == SOLVE:
            i = a
            assert k + 3 <= i < T
            need_sol(i, k, what)
            need_ver(i >> 1, k >> 1, k - 1, what)
            assert sol[i] == k
            sol[i] = k + 1
            solved.add((i, k))
        elif typ == PAIR_PREP:                      # pair inverse / coupling for the paired back substitution: a = pair, b = part
            npairs = (T - 3) // 2 if T >= 7 else 0
            gp, part = a, b
            lo, hi = 2 * gp, 2 * gp + 1
            assert 0 <= gp < npairs and 0 <= part < (3 if gp + 1 < npairs else 1)
            need_sol(hi, lo + 1, what)              # tile (hi, lo)
            if gp + 1 < npairs:
                need_sol(lo + 2, lo + 2, what); need_sol(lo + 3, lo + 2, what)      # tiles (lo+2, {lo, hi}), (lo+3, {lo, hi})
        else:
            assert typ == UPDATE and k >= 1
            I, J, part, parts, target = a, b & 255, (b >> 8) & 15, (b >> 12) & 15, b >> 16
            need_ver(I, J, k - 1, what)
            npan = parts - 6 if parts > 6 else 1   # np > 1: the whole super-tile by the panels k-1 .. k+np-2 (far from the front)
            assert npan in (1, 2, 3, 4)
            if parts == 1 or npan > 1:
                rows = [2 * I, 2 * I + 1] + ([2 * J, 2 * J + 1] if J != I else [])
                tiles = [(2 * I + (q >> 1), 2 * J + (q & 1)) for q in range(4)]
            elif parts == 2:
                bi = 2 * I + part
                rows = [bi] + [r for r in (2 * J, 2 * J + 1) if r != bi]
                tiles = [(bi, 2 * J), (bi, 2 * J + 1)]
            else:
                assert parts == 4
                bi, bj = 2 * I + (part >> 1), 2 * J + (part & 1)
                rows = [bi] + ([bj] if bj != bi else [])
                tiles = [(bi, bj)]
            for r in rows:
                if r < T and r >= k + 1:
                    need_sol(r, k + npan - 1, what)
            for (r, c) in tiles:
                if _valid(T, k, r, c):
                    assert applied.get((r, c), []) == list(range(0, k - 1)), what + ": panels applied to tile (%d,%d) out of order" % (r, c)
                    applied.setdefault((r, c), []).append(k - 1)
                    for kk in range(k + 1, k + npan):
                        assert _valid(T, kk, r, c), what + ": tile (%d,%d) does not take panel %d" % (r, c, kk - 1)
                        applied[(r, c)].append(kk - 1)
                else:
                    for kk in range(k + 1, k + npan):
                        assert not _valid(T, kk, r, c)
            sub[(I, J)] = sub.get((I, J), 0) + 1
            assert sub[(I, J)] <= target
            if sub[(I, J)] == target:
                ver[(I, J)] = k + npan - 1
    # completeness: every tile below the first sub-diagonal is solved by a task; every trailing tile got every panel it needs from an
    # update task (the panels k-1 of the chain's / prep's own three tiles are applied by those themselves)
    for c in range(T - 1):
        for r in range(c + 1, T):
            if r == c + 1:
                assert (r, c) in solved or c + 2 >= T      # the chain's X tile: copied by PrepX(c), the last one stored by the chain
            else:
                assert (r, c) in solved, "tile (%d,%d) is never solved" % (r, c)
    for c in range(1, T):
        for r in range(c, T):
            want = [p for p in range(0, c) if _valid(T, p + 1, r, c)]
            assert applied.get((r, c), []) == want, "tile (%d,%d): panels %s applied, %s expected" % (r, c, applied.get((r, c), []), want)


def _sparse_list(T, nz):
    L = _capi.lib()
    n = C.c_int64(0)
    nz = np.ascontiguousarray(nz, dtype=np.uint8)
    m = np.zeros((T, T), dtype=np.uint8)
    u8 = lambda a: a.ctypes.data_as(C.POINTER(C.c_uint8))
    assert L.pp_cholesky_task_list_sparse(T, u8(nz), u8(m), None, 0, C.byref(n)) == 0
    buf = np.zeros(7 * n.value, dtype=np.int32)
    assert L.pp_cholesky_task_list_sparse(T, u8(nz), u8(m), buf.ctypes.data_as(C.POINTER(C.c_int32)), n.value, C.byref(n)) == 0
    return buf.reshape(-1, 7), m.astype(bool)


def _band(T, w):
    return np.tril(np.ones((T, T), dtype=np.uint8)) * (np.subtract.outer(np.arange(T), np.arange(T)) <= w)


def _arrow(T, w, border):      # a band plus dense last rows (a loop closure / the right-hand side's row)
    m = _band(T, w)
    m[T - border:, :] = 1
    return np.tril(m)


def _dissected(T, w, parts):   # nested-dissection shape: independent diagonal bands, separators ordered last
    m = np.zeros((T, T), dtype=np.uint8)
    nsep = w * (parts - 1)
    size = (T - nsep) // parts
    for p in range(parts):
        lo, hi = p * size, (p + 1) * size if p + 1 < parts else T - nsep
        m[lo:hi, lo:hi] = _band(hi - lo, w)
        m[T - nsep:, lo:hi] = 1
    m[T - nsep:, T - nsep:] = 1
    return np.tril(m)


@pytest.mark.parametrize("T,shape", [(8, "band1"), (24, "band1"), (47, "band4"), (47, "band9"), (48, "arrow"), (64, "arrow"), (47, "dense"), (47, "dissected"),
                                     (94, "dissected"), (128, "band6")])
def test_sparse_task_list_waits_only_for_earlier_tasks_and_covers_the_structure(T, shape):
    """The one-launch factorisation of a BLOCK-SPARSE system: tasks exist only for the structurally non-zero tiles, and the counter values a task
    waits for (ChainTask::w0, w1, w2) are those an EXISTING earlier task produces.  Replay: every wait is satisfied by an earlier task of the
    list, the solves of a row happen in column order, every non-zero tile is solved, and every trailing tile receives exactly the panels that
    couple it (in increasing order).  A dense map gives the dense list's waits (k - 1 / k)."""
    nz = {"band1": lambda: _band(T, 1), "band4": lambda: _band(T, 4), "band9": lambda: _band(T, 9), "band6": lambda: _band(T, 6), "arrow": lambda: _arrow(T, 3, 2),
          "dense": lambda: np.tril(np.ones((T, T), dtype=np.uint8)), "dissected": lambda: _dissected(T, 3, 4)}[shape]()
    tasks, has = _sparse_list(T, nz)
    assert has[np.tril_indices(T)][np.asarray(np.tril(nz))[np.tril_indices(T)] > 0].all()      # the map contains the input
    for k in range(T):
        assert has[k, k] and (k + 1 >= T or has[k + 1, k]) and (k + 2 >= T or has[k + 2, k])     # ... and the two sub-diagonals
    sol = np.zeros(T + 4, dtype=np.int64)
    ver, sub, applied, solved = {}, {}, {}, set()

    def couples(r, c, p):
        return has[r, p] and has[c, p]

    for typ, k, a, b, w0, w1, w2 in tasks:
        what = "task (type %d, k %d, a %d, b 0x%x)" % (typ, k, a, b)
        if typ in (PREP_X, PREP_D):
            assert k + 2 < T
            if k > 0:
                assert ver.get(((k + 2) >> 1, k >> 1), 0) >= w0, what
                assert ver.get(((k + 2) >> 1, ((k + 1) >> 1) if typ == PREP_X else ((k + 2) >> 1)), 0) >= w1, what
                far = bool(has[k + 2, k - 1])
                assert sol[k + 2] >= (w2 if typ == PREP_X else (k if far else 0)), what
                if typ == PREP_X:
                    assert w2 >= (k if far else 0) and sol[k + 1] >= k, what
                # every panel that couples one of the task's tiles has been applied to its super-tile (its own panel k-1 it applies itself)
                for (r, c) in ((k + 2, k), (k + 2, k + 1) if typ == PREP_X else (k + 2, k + 2)):
                    want = [p for p in range(0, k - 1) if couples(r, c, p)]
                    assert applied.get((r, c), []) == want, what + ": tile (%d,%d) has panels %s, needs %s" % (r, c, applied.get((r, c), []), want)
            if typ == PREP_X:
                assert sol[k + 2] <= k + 1 and sol[k + 1] <= k + 1
                sol[k + 2] = k + 1; sol[k + 1] = k + 1
                solved.add((k + 2, k)); solved.add((k + 1, k))
        elif typ == SOLVE:
            i = a
            assert k + 3 <= i < T and has[i, k], what
            assert sol[i] >= w2 and ver.get((i >> 1, k >> 1), 0) >= w0, what
            prev = [c for c in range(k) if has[i, c]]
            assert w2 == (prev[-1] + 1 if prev else 0), what + ": the row's previous non-zero column"
            assert all((i, c) in solved for c in prev), what + ": an earlier column of the row is still unsolved"
            want = [p for p in range(0, k - 1) if couples(i, k, p)]
            assert applied.get((i, k), []) == want, what + ": tile has panels %s, needs %s" % (applied.get((i, k), []), want)
            sol[i] = k + 1
            solved.add((i, k))
        else:
            assert typ == UPDATE and k >= 1
            I, J, part, parts, target = a, b & 255, (b >> 8) & 15, (b >> 12) & 15, b >> 16
            assert parts in (1, 2, 4), what + ": block-sparse lists hold single-panel updates"
            assert ver.get((I, J), 0) >= w0, what
            if parts == 1:
                tiles = [(2 * I + (q >> 1), 2 * J + (q & 1)) for q in range(4)]
            elif parts == 2:
                tiles = [(2 * I + part, 2 * J), (2 * I + part, 2 * J + 1)]
            else:
                tiles = [(2 * I + (part >> 1), 2 * J + (part & 1))]
            for (r, c) in tiles:
                if _valid(T, k, r, c) and couples(r, c, k - 1):
                    assert (r, k - 1) in solved and (c, k - 1) in solved, what + ": an operand of tile (%d,%d) is unsolved" % (r, c)
                    want = [p for p in range(0, k - 1) if couples(r, c, p) and _valid(T, p + 1, r, c)]
                    assert applied.get((r, c), []) == want, what + ": panels applied to tile (%d,%d) out of order" % (r, c)
                    applied.setdefault((r, c), []).append(k - 1)
            sub[(I, J)] = sub.get((I, J), 0) + 1
            assert sub[(I, J)] <= target
            if sub[(I, J)] == target:
                ver[(I, J)] = k
    for c in range(T - 1):
        for r in range(c + 1, T):
            if has[r, c] and not (r == c + 1 and c + 2 >= T):
                assert (r, c) in solved, "tile (%d,%d) is never solved" % (r, c)
    for c in range(1, T):
        for r in range(c, T):
            want = [p for p in range(0, c) if _valid(T, p + 1, r, c) and couples(r, c, p)]
            assert applied.get((r, c), []) == want, "tile (%d,%d): panels %s applied, %s expected" % (r, c, applied.get((r, c), []), want)
    if shape == "dense":      # the waits of the dense list's device code: k - 1 panels applied, column k - 1 solved
        for typ, k, a, b, w0, w1, w2 in tasks:
            if typ in (PREP_X, PREP_D) and k > 0:
                assert w0 == k - 1 and w1 == k - 1 and (typ == PREP_D or w2 == k)
            if typ == SOLVE:
                assert w0 == max(k - 1, 0) and w2 == k      # (a target <= 0 is no wait)
            if typ == UPDATE:
                assert w0 == k - 1


# ---- several chains: a nested-dissection order factorises the independent sub-trees of the elimination tree side by side -------------------------------
def _plan(T, nz, max_chains=0):
    L = _capi.lib()
    n = C.c_int64(0)
    nz = np.ascontiguousarray(nz, dtype=np.uint8)
    m = np.zeros((T, T), dtype=np.uint8)
    chains = np.zeros(49, dtype=np.int32)
    time = np.zeros(T, dtype=np.int32)
    rho1 = np.zeros(T, dtype=np.int32)
    ok = C.c_int32(0)
    u8 = lambda a: a.ctypes.data_as(C.POINTER(C.c_uint8))
    i32 = lambda a: a.ctypes.data_as(C.POINTER(C.c_int32))
    assert L.pp_cholesky_task_plan(T, u8(nz), max_chains, u8(m), None, 0, C.byref(n), i32(chains), i32(time), i32(rho1), C.byref(ok)) == 0
    buf = np.zeros(16 * n.value, dtype=np.int32)
    assert L.pp_cholesky_task_plan(T, u8(nz), max_chains, u8(m), i32(buf), n.value, C.byref(n), i32(chains), i32(time), i32(rho1), C.byref(ok)) == 0
    ranges = [(int(chains[1 + 3 * c]), int(chains[2 + 3 * c]), int(chains[3 + 3 * c])) for c in range(int(chains[0]))]
    return buf.reshape(-1, 16), m.astype(bool), ranges, time, rho1, bool(ok.value)


def _leaves(T, sizes, w, nsep):      # independent diagonal bands of the given sizes (block columns), then `nsep` separator block columns coupled to everything
    m = np.zeros((T, T), dtype=np.uint8)
    lo = 0
    for size in sizes:
        m[lo:lo + size, lo:lo + size] = _band(size, w)
        lo += size
    assert lo + nsep == T
    m[lo:, :] = 1
    return np.tril(m)


def _two_level(T, leaf, w, sep1, top):      # [leaf leaf sep1] [leaf leaf sep1] top: sep1 couples its two leaves, top couples everything
    m = np.zeros((T, T), dtype=np.uint8)
    lo = 0
    for half in range(2):
        h0 = lo
        for _ in range(2):
            m[lo:lo + leaf, lo:lo + leaf] = _band(leaf, w)
            lo += leaf
        m[lo:lo + sep1, h0:lo + sep1] = 1
        lo += sep1
    assert lo + top == T
    m[lo:, :] = 1
    return np.tril(m)


MERGE = 6
C_SOL0, MAX_STEPS, MAX_SUPER = 8, 128, 65
C_VER0 = C_SOL0 + 16 * MAX_STEPS
C_SUB0 = C_VER0 + MAX_SUPER * MAX_SUPER


def _replay_plan(T, tasks, has, ranges, time, rho1):
    """the device's waits and stores on its counters (by absolute index, as the tasks carry them), the panels every tile has received - in the tile itself
    or in a chain's scratch array until that chain's merge task adds it - and what is solved"""
    chain_of = np.zeros(T, dtype=np.int64)
    for c, (b, e, _) in enumerate(ranges):
        chain_of[b:e] = c
    begin = lambda k: ranges[chain_of[k]][0]
    end = lambda k: ranges[chain_of[k]][1]
    ctr = {}
    ver = lambda I, J: ctr.get(C_VER0 + I * MAX_SUPER + J, 0)
    sol = lambda c, row: ctr.get(C_SOL0 + c * MAX_STEPS + row, 0)
    applied, zapplied, solved = {}, {}, set()
    px, pd = set(), set()

    def can_run(s):      # chain step s: the solve of tile (s+1,s), M_(s+1)
        if s + 1 >= end(s):
            return False
        return all((q - 1) in px and (q - 1) in pd for q in range(begin(s) + 1, s + 1))

    def couples(r, c, p):
        return p < c and has[r, p] and has[c, p]

    def coupling(r, c, but=()):
        return {p for p in range(c) if couples(r, c, p)} - set(but)

    def chain_stores(row, col):      # the last solved tile of a chain that stops: stored (and its row counter moved) by the chain itself
        b, e, post = ranges[chain_of[col]]
        return e < T and row == e - 1 and col == e - 2 and can_run(col)

    def post(idx, value, what):
        assert ctr.get(idx, 0) < value, what + ": a counter would move backwards"
        ctr[idx] = value

    zslot, slot_owner = {}, {}
    for typ, k, a, b, w0, w1, w2, flags, cidx, sidx, zsel, mask, s0, s1, s2, s3 in tasks:
        slots = (s0, s1, s2, s3)
        what = "task (type %d, k %d, a %d, b 0x%x)" % (typ, k, a, b)
        first = bool(flags & 1)
        fc = (flags >> 4) & 15
        if typ in (PREP_X, PREP_D):
            assert fc == chain_of[k] and first == (k == begin(k)) and k + 2 < end(k), what
            X = typ == PREP_X
            out = (k + 2, k + 1) if X else (k + 2, k + 2)
            if not first:
                assert b == rho1[k], what
                assert ver((k + 2) >> 1, k >> 1) >= w0 and ver((k + 2) >> 1, out[1] >> 1) >= w1, what
                far = bool(has[k + 2, k - 1])
                assert sol(fc, k + 2) >= (w2 if X else (a if far else 0)), what
                if X:
                    assert sol(fc, k + 1) >= a and (k + 1, k - 1) in solved, what
                assert can_run(k - 1), what + ": M_k / the solved tile (k,k-1) cannot exist yet"
                assert not far or (k + 2, k - 1) in solved, what
            for (r, c) in ((k + 2, k), out):      # what the update tasks have applied; the pending panel k-1 (and k, to the output tile) it applies itself
                want = {p for p in coupling(r, c) if p < k - 1}      # (the panels k+1.. of the output tile come later: the chain's)
                assert applied.get((r, c), set()) == want, what + ": tile (%d,%d) has panels %s, needs %s" % (r, c, sorted(applied.get((r, c), set())), sorted(want))
                if first:
                    assert not {p for p in coupling(r, c) if p < k}, what
            if X:
                assert can_run(k), what + ": the solved tile (k+1,k) cannot exist yet"
                post(C_SOL0 + fc * MAX_STEPS + k + 2, b, what)
                post(C_SOL0 + fc * MAX_STEPS + k + 1, b, what)
                solved.add((k + 2, k)); solved.add((k + 1, k))
                px.add(k)
            else:
                pd.add(k)
        elif typ == SOLVE:
            i = a
            assert fc == chain_of[k] and first == (k == begin(k)) and k + 3 <= i < T and has[i, k], what
            assert sol(fc, i) >= w2 and ver(i >> 1, k >> 1) >= w0, what
            if not first:
                assert can_run(k - 1), what
                assert not has[i, k - 1] or (i, k - 1) in solved, what + ": tile (i,k-1) of the pending panel is unsolved"
            else:
                assert not coupling(i, k), what
            want = coupling(i, k, but=(k - 1,))
            assert applied.get((i, k), set()) == want, what + ": tile has panels %s, needs %s" % (sorted(applied.get((i, k), set())), sorted(want))
            assert w1 == rho1[k], what
            post(C_SOL0 + fc * MAX_STEPS + i, w1, what)
            solved.add((i, k))
        elif typ == MERGE:
            I, J, c = a, b & 255, zsel
            assert 0 <= c < len(ranges) and cidx == C_VER0 + I * MAX_SUPER + J, what
            assert ctr.get(cidx, 0) >= w0 and ctr.get(sidx, 0) >= w2, what
            for q in range(4):
                tile = (2 * I + (q >> 1), 2 * J + (q & 1))
                z = zapplied.pop((c, tile), set())
                assert bool((mask >> q) & 1) == bool(z), what + ": merges exactly the tiles the chain accumulated for"
                assert not z or slots[q] == zslot[(c, tile)], what + ": ... from the scratch tile they were accumulated in"
                assert not (applied.get(tile, set()) & z), what
                applied.setdefault(tile, set()).update(z)
            post(cidx, w1, what)
        else:
            assert typ == UPDATE and k >= 1
            I, J, part, parts, target = a, b & 255, (b >> 8) & 15, (b >> 12) & 15, b >> 16
            assert parts in (1, 2, 4), what + ": block-sparse lists hold single-panel updates"
            assert fc == chain_of[k - 1] and zsel in (-1, fc) and w2 == rho1[k - 1], what
            if zsel < 0:
                assert cidx == C_VER0 + I * MAX_SUPER + J and sidx == C_SUB0 + I * MAX_SUPER + J, what
            assert ctr.get(cidx, 0) >= w0, what
            if parts == 1:
                tiles = [(q, 2 * I + (q >> 1), 2 * J + (q & 1)) for q in range(4)]
                rows = {2 * I, 2 * I + 1, 2 * J, 2 * J + 1}
            elif parts == 2:
                tiles = [(2 * part, 2 * I + part, 2 * J), (2 * part + 1, 2 * I + part, 2 * J + 1)]
                rows = {2 * I + part, 2 * J, 2 * J + 1}
            else:
                tiles = [(part, 2 * I + (part >> 1), 2 * J + (part & 1))]
                rows = {tiles[0][1], tiles[0][2]}
            for row in rows:      # the device's row_slot: waited for when the row has a tile in column k-1
                if row < T and row >= k + 1 and has[row, k - 1]:
                    have = max(sol(fc, row), ranges[fc][2] if chain_stores(row, k - 1) else 0)
                    assert have >= w2, what + ": row %d" % row
            for (q, r, c) in tiles:
                if _valid(T, k, r, c) and couples(r, c, k - 1):
                    for row in (r, c):
                        assert (row, k - 1) in solved or chain_stores(row, k - 1), what + ": an operand of tile (%d,%d) is unsolved" % (r, c)
                    assert (zsel < 0) == (chain_of[k - 1] == chain_of[c]), what + ": a panel goes to the tile itself exactly when its chain owns the tile's block column"
                    dst = applied.setdefault((r, c), set()) if zsel < 0 else zapplied.setdefault((zsel, (r, c)), set())
                    if zsel >= 0:
                        assert bool((mask >> q) & 1) == (not dst), what + ": taken as zero exactly when nothing has been accumulated yet"
                        assert slots[q] >= 0 and zslot.setdefault((zsel, (r, c)), slots[q]) == slots[q], what + ": one scratch tile per (chain, tile)"
                        assert slot_owner.setdefault(slots[q], (zsel, (r, c))) == (zsel, (r, c)), what + ": ... and nobody else's"
                    assert (k - 1) not in dst, what
                    dst.add(k - 1)
            ctr[sidx] = ctr.get(sidx, 0) + 1
            assert ctr[sidx] <= target
            if ctr[sidx] == target:
                post(cidx, w1, what)
    # every non-zero tile below the diagonal is solved (the chain's own tiles (k+1,k): by can_run), every tile got the panels that couple it
    assert not zapplied, "something is left in a scratch array"
    for c in range(T - 1):
        for r in range(c + 1, T):
            if not has[r, c]:
                continue
            if r == c + 1 and r < end(c):
                assert (r, c) in solved or (c + 2 >= end(c) and can_run(c)), "the chain's tile (%d,%d) is never solved" % (r, c)
            else:
                assert (r, c) in solved, "tile (%d,%d) is never solved" % (r, c)
    for c in range(1, T):
        for r in range(c, T):
            if not has[r, c]:
                continue
            own = {c - 1} | ({c - 2} if r <= c + 1 else set()) | ({c - 3} if r == c else set())
            want = coupling(r, c, but=own)
            assert applied.get((r, c), set()) == want, "tile (%d,%d): panels %s applied, %s expected" % (r, c, sorted(applied.get((r, c), set())), sorted(want))


@pytest.mark.parametrize("shape", ["two_leaves", "uneven_leaves", "four_leaves", "two_level", "odd_boundaries", "band", "dissected_unaligned"])
def test_task_plan_of_several_chains_is_a_topological_order_and_complete(shape):
    """A nested-dissection order puts independent sub-trees of the elimination tree side by side: every leaf gets a chain workgroup of its own, the
    block columns are eliminated in the order of their depth, and every counter value comes from the host (ChainTask a / b / w0 / w1 / w2).  Replay: the
    waits are met by earlier tasks, counters only grow, every tile is solved and receives exactly the panels that couple it before it is used."""
    T, nz, want_chains = {
        "two_leaves": lambda: (47, _leaves(47, [21, 21], 3, 5), 2),
        "uneven_leaves": lambda: (47, _leaves(47, [9, 30], 4, 8), 2),
        "four_leaves": lambda: (47, _leaves(47, [9, 9, 9, 9], 2, 11), 4),
        "two_level": lambda: (47, _two_level(47, 6, 2, 5, 13), 4),
        "odd_boundaries": lambda: (40, _leaves(40, [7, 11, 13], 3, 9), 3),
        "band": lambda: (47, _band(47, 4), 1),
        "dissected_unaligned": lambda: (47, _dissected(47, 3, 4), None),
    }[shape]()
    tasks, has, ranges, time, rho1, ok = _plan(T, nz)
    assert ok, "the host replay of the waits (TaskListWaitsAreMet) rejects its own list"
    if want_chains is not None:
        assert len(ranges) == want_chains, ranges
    assert has[np.tril_indices(T)][np.asarray(np.tril(nz))[np.tril_indices(T)] > 0].all()
    assert max(time) + 1 <= T and (len(ranges) == 1 or max(time) + 1 < T)
    _replay_plan(T, tasks, has, ranges, time, rho1)
    # one chain forced: the list of the single-chain path
    tasks1, has1, ranges1, time1, rho11, ok1 = _plan(T, nz, max_chains=1)
    assert ok1 and len(ranges1) == 1 and list(time1) == list(range(T)) and list(rho11) == list(range(1, T + 1))
    _replay_plan(T, tasks1, has1, ranges1, time1, rho11)


def _random_forest(rng, T):
    """random block structure: independent banded leaves, then separator rows that couple random subsets of the earlier columns, a few stray couplings"""
    nleaf = int(rng.integers(1, 7))
    nsep = int(rng.integers(1, max(2, T // 4)))
    if T - nsep - 4 * nleaf < 0:
        nleaf = 1
    sizes = rng.multinomial(T - nsep - 4 * nleaf, np.ones(nleaf) / nleaf) + 4
    m = np.zeros((T, T), dtype=np.uint8)
    lo = 0
    for s in sizes:
        m[lo:lo + s, lo:lo + s] = _band(int(s), int(rng.integers(1, 6)))
        lo += int(s)
    for r in range(lo, T):
        m[r, :r] = rng.uniform(size=r) < rng.uniform(0.1, 0.9)
        m[r, r] = 1
    m[T - 1, :] = 1
    for _ in range(int(rng.integers(0, 4))):
        i = int(rng.integers(1, T)); j = int(rng.integers(0, i)); m[i, j] = 1
    return np.tril(m)


@pytest.mark.parametrize("seed", range(8))
def test_task_plans_of_random_structures(seed):
    """random forests of banded leaves under random separators (and stray couplings that merge chains): whatever chains the plan finds, the host replay of
    the library accepts its list and the replay here - waits, counters, scratch arrays, merges, coverage - agrees"""
    rng = np.random.default_rng(1000 + seed)
    multi = 0
    for _ in range(12):
        T = int(rng.integers(12, 100))
        nz = _random_forest(rng, T)
        tasks, has, ranges, time, rho1, ok = _plan(T, nz)
        assert ok, (T, ranges)
        multi += len(ranges) > 1
        _replay_plan(T, tasks, has, ranges, time, rho1)
    assert multi >= 4


def test_a_plan_that_runs_out_of_scratch_counters_says_so():
    """eight leaves under a separator of 48 block columns: every leaf accumulates for ~300 separator super-tiles, more scratch sequences than the counter pool
    holds - the plan is reported as unusable (pp_ba_solve / pp_dense_cholesky_solve then take ONE chain: PlanAndList), not silently truncated"""
    T = 128
    nz = _leaves(T, [10] * 8, 2, 48)
    tasks, has, ranges, time, rho1, ok = _plan(T, nz)
    assert len(ranges) == 8 and not ok
    tasks1, has1, ranges1, time1, rho11, ok1 = _plan(T, nz, max_chains=1)
    assert len(ranges1) == 1 and ok1
    _replay_plan(T, tasks1, has1, ranges1, time1, rho11)
