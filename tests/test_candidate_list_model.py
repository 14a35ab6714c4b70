"""CPU model of the Harvest candidate LISTS against the oracle's dense maps.

The reference keeps the refined candidates of a 1 ms frame in a [7 * max_candidates] column, zeros where there is none
(world/harvest.py:114-125, 215-234).  The HIP path stores, per frame, the attempted candidates in row order (a list), a
keep mask over the list from the pruning pass, and nothing else.  The claim the kernels rest on is that the contour
stage only ever asks three questions of a column, and that each has the same answer on the list:

  * np.argmax of the pruned scores, then the candidate at that row (harvest.py:303)   -> first maximum in row order;
  * SelectBestF0 over the pruned candidates (harvest.py:238-248)                      -> last minimum in row order;
  * the best score among the rows whose candidate equals a given f0 (harvest.py:490-495) -> order-free.

This test restates the three on lists in NumPy and checks them against the oracle's own functions on the oracle's
maps for the golden utterances; no GPU involved.  (The GPU suite checks the kernels end to end.)"""
import numpy as np
import pytest


def _lists(refined_f0, pruned_f0, pruned_sc, attempted):
    """Per frame: (values, scores, keep) in row order over the rows that held a candidate before refinement."""
    out = []
    for j in range(refined_f0.shape[1]):
        rows = np.nonzero(attempted[:, j] != 0)[0]
        vals = refined_f0[rows, j]
        keep = pruned_f0[rows, j] != 0
        sc = np.where(keep, pruned_sc[rows, j], 0.0)
        out.append((vals, sc, keep))
    return out


def _argmax_list(vals, sc, keep):
    best, bs = -1, 0.0
    for k in range(len(vals)):
        se = sc[k] if keep[k] else 0.0
        if se > bs:
            bs, best = se, k
    return vals[best] if best >= 0 else 0.0


def _pick_list(ref, vals, keep, allowed):
    best, best_err = 0.0, allowed
    for k in range(len(vals)):
        c = vals[k] if keep[k] else 0.0
        e = abs(ref - c) / ref
        if e > best_err:
            continue
        best, best_err = c, e
    return best


def _score_list(f0, vals, sc, keep):
    s = 0
    if f0 == 0:
        return s
    for k in range(len(vals)):
        if keep[k] and f0 == vals[k] and s < sc[k]:
            s = sc[k]
    return s


@pytest.mark.parametrize("tag", ["syn16k"])
def test_list_semantics_equal_dense_map(golden, tag):
    from oracle import pitch_harvest as ph

    g = golden(tag)
    fs = int(g["fs"])
    x = g["x"][: int(1.2 * fs)]
    aux = ph.harvest_np(x, fs, return_aux=True)["aux"]
    cf0, pf0, psc, att = aux["refined_f0"], aux["pruned_f0"], aux["pruned_score"], aux["overlapped"]
    assert cf0.shape == pf0.shape == psc.shape == att.shape
    # what the layout argument needs of the data: a surviving candidate has a positive score, a row without a candidate
    # never survives, and a refined zero is a zero in the pruned map
    assert np.all(psc[pf0 != 0] > 0)
    assert np.all(pf0[att == 0] == 0) and np.all(pf0[cf0 == 0] == 0)
    lists = _lists(cf0, pf0, psc, att)
    assert max(len(v) for v, _, _ in lists) <= 105
    nfr = pf0.shape[1]
    # 1. base contour
    dense_base = pf0[np.argmax(psc, axis=0), np.arange(nfr)]
    list_base = np.array([_argmax_list(*lists[j]) for j in range(nfr)])
    assert np.array_equal(dense_base, list_base)
    # 2. SelectBestF0 with the contour stage's allowed range, on references near and away from the candidates
    rng = np.random.RandomState(3)
    frames = rng.choice(nfr, size=300, replace=False)
    for j in frames:
        vals, sc, keep = lists[j]
        col = pf0[:, j]
        refs = [float(r) for r in col[col != 0][:3]] + [float(r) * 1.1 for r in col[col != 0][:2]] + [150.0, 431.7]
        for ref in refs:
            assert ph._pick(ref, col, 0.18) == _pick_list(ref, vals, keep, 0.18)
            # 3. score look-up by value
            assert ph._score_of(ref, col, psc[:, j]) == _score_list(ref, vals, sc, keep)


def test_ties_resolve_by_row_order():
    """Equal scores: the earlier row wins the argmax; equal errors: the later row wins SelectBestF0 — on lists as on maps."""
    from oracle import pitch_harvest as ph

    col = np.zeros(105)
    sc = np.zeros(105)
    col[[4, 19, 40, 77]] = [100.0, 120.0, 80.0, 120.0]
    sc[[4, 19, 40, 77]] = [7.0, 9.0, 9.0, 3.0]
    rows = np.array([4, 19, 33, 40, 77])  # row 33 was attempted and refined to zero
    vals, scs, keep = col[rows], sc[rows], col[rows] != 0
    assert col[np.argmax(sc)] == _argmax_list(vals, scs, keep) == 120.0
    assert ph._pick(100.0, col, 0.25) == _pick_list(100.0, vals, keep, 0.25) == 100.0
    assert ph._pick(100.0, col[[19, 40]], 0.25) == 80.0  # |100-120| == |100-80|: the later row
    assert _pick_list(100.0, col[[19, 40]], np.array([True, True]), 0.25) == 80.0
    assert ph._score_of(120.0, col, sc) == _score_list(120.0, vals, scs, keep) == 9.0
