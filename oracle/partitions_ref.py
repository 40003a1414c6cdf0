"""oracle/partitions_ref.py -- the perturbation partitions of reference kalman.py, statement by statement.

TEST INFRASTRUCTURE ONLY (tests/ and tools/make_golden.py import it; the product never does).

Restates, in the reference's own order of operations:
  * the Jacobian partitions ``E`` and their triangle labels ``labels``  (kalman.py:223-272);
  * the Hessian pair list ``Q``, its partitions ``E_hessian`` / ``E_hessian_idx`` and
    ``labels_hess``                                                      (kalman.py:305-389).
The reference uses ``union2d / intersect2d / setdiff2d`` from the author's un-vendored ``useful``
module (kalman.py:21); they are restated here as what their names and call sites say: set operations
on the ROWS of 2-column integer arrays with sorted (lexicographic) unique rows as result, the 2-D
counterparts of numpy's union1d / intersect1d / setdiff1d that the 1-D index bookkeeping next to
them (``Qidx``, ``Pidx``, ``Aidx``) uses -- the pair list Q is generated in lexicographic order, so row
order and index order agree and the two bookkeepings stay in step, as the reference relies on.

Integer work: the product's KFState must reproduce these lists exactly
(tests/test_host_cpu.py::test_partitions_equal_reference_restatement, fixture
tests/golden/partitions_config1.npz).  Two properties of the reference worth knowing, both kept:
  * in the inner loop the "do later" set P is intersected with A *before* the current element is
    removed from A (kalman.py:239-240, 341-344), and for the pairs the current pair is itself part of
    p_all (p_self1 / p_self2 match it when it is a diagonal pair): a diagonal pair (i,i) is therefore
    scheduled again and appears in a second class -- the defect the author notes at
    testbites/test_multipert_validation.py:330-332;
  * labels are assigned member by member, later members overwriting earlier ones (kalman.py:263-272).
"""
import numpy as np


def _rows(a):
    a = np.asarray(a)
    if a.size == 0:
        return np.zeros((0, 2), np.int64)
    return a.reshape(-1, 2).astype(np.int64)


def _sorted_unique(rows):
    if len(rows) == 0:
        return np.zeros((0, 2), np.int64)
    return np.unique(rows, axis=0)


def union2d(a, b):
    return _sorted_unique(np.vstack((_rows(a), _rows(b))))


def intersect2d(a, b):
    sb = set(map(tuple, _rows(b)))
    return _sorted_unique(np.array([r for r in _rows(a) if tuple(r) in sb], np.int64).reshape(-1, 2))


def setdiff2d(a, b):
    sb = set(map(tuple, _rows(b)))
    return _sorted_unique(np.array([r for r in _rows(a) if tuple(r) not in sb], np.int64).reshape(-1, 2))


def adjacency(N, tri):
    """Jv, kalman.py:189-196."""
    Jv = np.eye(N)
    for t in tri:
        Jv[t[0], t[1]] = 1
        Jv[t[0], t[2]] = 1
        Jv[t[1], t[2]] = 1
        Jv[t[1], t[0]] = 1
        Jv[t[2], t[0]] = 1
        Jv[t[2], t[1]] = 1
    return Jv


def jacobian_partitions(N, tri):
    """kalman.py:223-272 -> (E: list of lists of vertex ids, labels: T x len(E))."""
    Jv = adjacency(N, tri)
    E = []
    Q = np.arange(N)
    A = np.arange(N)
    while len(Q) > 0:
        P = np.array([])
        e = []
        while len(Q) > 0:
            q = Q[0]
            p = np.nonzero(Jv[q, :])[0]
            p = np.setdiff1d(p, q)
            e += [int(q)]
            P = np.intersect1d(np.union1d(P, p), A)
            A = np.setdiff1d(A, q)
            Q = np.setdiff1d(Q, p)
            Q = np.setdiff1d(Q, q)
        Q = P.astype(np.int64)
        E += [e]
    labels = -1 * np.ones((len(tri), len(E)))
    for k, e in enumerate(E):
        label = -1 * np.ones(len(tri))
        for node in e:
            for j, t in enumerate(tri):
                if node in t:
                    label[j] = node
        labels[:, k] = label
    return E, labels


def hessian_partitions(N, tri):
    """kalman.py:305-389 -> (Q: pairs, E_hessian: list of (k,2) arrays, E_hessian_idx: list of index
    arrays into Q, labels_hess: T x len(E_hessian))."""
    Jv = adjacency(N, tri)
    E_hessian, E_hessian_idx = [], []
    Q = []
    for i in range(N):
        for j in range(i, N):
            if Jv[i, j]:
                Q = Q + [[i, j]]
    Q = np.array(Q)
    Qfull = Q.copy()
    Qidx = np.arange(len(Q))
    A = Q.copy()
    Aidx = Qidx.copy()
    while len(Q) > 0:
        P = np.array([])
        Pidx = np.array([])
        e = np.array([])
        eidx = np.array([])
        while len(Q) > 0:
            q = Q[0]
            qidx = Qidx[0]
            p1 = np.nonzero(Jv[q[0], :])[0]
            p2 = np.nonzero(Jv[q[1], :])[0]
            p = np.union1d(p1, p2)
            p = np.setdiff1d(p, q)
            p_all1 = np.array([i in p for i in Q[:, 0]])
            p_all2 = np.array([i in p for i in Q[:, 1]])
            p_self1 = np.all(Q == [q[0], q[0]], 1)
            p_self2 = np.all(Q == [q[1], q[1]], 1)
            p_all_idx = p_all1 | p_all2 | p_self1 | p_self2          # the reference adds the boolean arrays
            p_all = Q[p_all_idx, :]
            p_all_idx = Qidx[p_all_idx]
            e = union2d(e, q)
            eidx = np.union1d(eidx, [qidx])
            P = intersect2d(union2d(P, p_all), A)
            Pidx = np.intersect1d(np.union1d(Pidx, p_all_idx), Aidx)
            A = setdiff2d(A, q)
            Aidx = np.setdiff1d(Aidx, qidx)
            Q = setdiff2d(Q, p_all)
            Q = setdiff2d(Q, q)
            Qidx = np.setdiff1d(Qidx, p_all_idx)
            Qidx = np.setdiff1d(Qidx, qidx)
        Q = P
        Qidx = Pidx.astype(np.int64)
        if len(e.shape) == 1:
            e = np.reshape(e, (-1, 2))
        E_hessian += [e]
        E_hessian_idx += [eidx]
    labels_hess = -1 * np.ones((len(tri), len(E_hessian)))
    for k, e in enumerate(E_hessian):
        label = -1 * np.ones(len(tri))
        for i, nodes in enumerate(e):
            n1, n2 = nodes
            for j, t in enumerate(tri):
                if (n1 in t) or (n2 in t):
                    label[j] = E_hessian_idx[k][i]
        labels_hess[:, k] = label
    return Qfull, E_hessian, E_hessian_idx, labels_hess
