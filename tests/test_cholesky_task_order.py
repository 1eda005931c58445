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
