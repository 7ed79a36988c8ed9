"""Device evaluation metrics (csrc/metrics.hip through the C ABI) vs the CPU oracle (oracle/metrics_oracle.py, pinned
to sklearn in tests/test_evaluation.py).  Integer state (positives, histograms) is bit-exact; AP values are f64 sums of
the same terms - tolerance 1e-12 relative."""
import numpy as np
import pytest
import torch

from oracle import metrics_oracle as MO

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _case(n, nl, seed, ties=False, prevalence=0.05, saturate=False):
    g = torch.Generator().manual_seed(seed)
    logits = torch.randn(n, nl, generator=g) * (8.0 if saturate else 2.0)
    y = (torch.rand(n, nl, generator=g) < prevalence)
    y = y | ((logits > 2.5) & (torch.rand(n, nl, generator=g) < 0.5))  # informative
    p = torch.sigmoid(logits)
    if ties:
        p = (p * 20).round() / 20
    return p.float(), y


def _oracle(p, y):
    p, y = p.numpy(), y.numpy()
    per = np.array([MO.average_precision_fast(p[:, j], y[:, j]) for j in range(p.shape[1])])
    return per, MO.average_precision_fast(p.ravel(), y.ravel())


@pytest.mark.parametrize("n,nl,batch,cap,kw", [
    (37, 5, 37, 37, {}),                              # one ragged tile
    (300, 70, 64, 300, dict(ties=True)),              # heavy ties, several appends
    (1000, 33, 256, 1200, {}),                        # capacity > n: compaction path for micro
    (9000, 12, 4096, 9000, dict(saturate=True)),      # several scan tiles per chunk, saturated sigmoid ties (0.0 / 1.0)
    (70000, 3, 8192, 70000, dict(ties=True)),         # several chunks per label, tie groups crossing chunk borders
])
@pytest.mark.parametrize("label_dtype", [torch.int64, torch.float32, torch.bool])
def test_exact_ap_matches_oracle(n, nl, batch, cap, kw, label_dtype):
    from protnote_amd.utils.evaluation import DeviceAveragePrecision

    p, y = _case(n, nl, seed=n + nl, **kw)
    if nl >= 5:
        y[:, 2] = False                               # label without positives -> NaN, skipped by the macro mean
        y[:, 4] = True                                # label with only positives -> AP 1
    acc = DeviceAveragePrecision(nl, cap, DEV)
    for i in range(0, n, batch):
        acc.update(p[i:i + batch].to(DEV), y[i:i + batch].to(DEV).to(label_dtype))
    out = acc.compute()
    per, micro = _oracle(p, y)
    got = out["ap_per_label"].cpu().numpy()
    assert np.array_equal(np.isnan(got), np.isnan(per))
    np.testing.assert_allclose(got[~np.isnan(per)], per[~np.isnan(per)], rtol=1e-12, atol=0)
    np.testing.assert_array_equal(out["positives_per_label"].cpu().numpy(), y.numpy().sum(0))
    np.testing.assert_allclose(out["map_micro"], micro, rtol=1e-12)
    np.testing.assert_allclose(out["map_macro"], MO.macro_mean(per), rtol=1e-12)
    if nl >= 5:
        assert got[4] == 1.0
    # computing twice (the accumulator is not consumed) and after a reset + refill gives the same bits
    again = acc.compute()
    assert torch.equal(again["ap_per_label"].nan_to_num(-1), out["ap_per_label"].nan_to_num(-1))
    assert again["map_micro"] == out["map_micro"]


def test_exact_ap_definition_small():
    """Against the O(n * thresholds) by-definition oracle (not the sort-based shortcut) and sklearn."""
    from sklearn.metrics import average_precision_score
    from protnote_amd.utils.evaluation import DeviceAveragePrecision

    p, y = _case(120, 6, seed=5, ties=True)
    acc = DeviceAveragePrecision(6, 120, DEV)
    acc.update(p.to(DEV), y.to(DEV))
    got = acc.compute()["ap_per_label"].cpu().numpy()
    for j in range(6):
        np.testing.assert_allclose(got[j], MO.average_precision(p[:, j].numpy(), y[:, j].numpy()), rtol=1e-12)
        np.testing.assert_allclose(got[j], average_precision_score(y[:, j].numpy(), p[:, j].numpy()), rtol=1e-12)


def test_exact_ap_order_properties():
    """Size-independent properties: AP is invariant under a permutation of the proteins and under a strictly
    increasing map of the scores; negative zero ties with zero."""
    from protnote_amd.utils.evaluation import DeviceAveragePrecision

    p, y = _case(5000, 40, seed=9)
    a = DeviceAveragePrecision(40, 5000, DEV)
    a.update(p.to(DEV), y.to(DEV))
    base = a.compute()
    perm = torch.randperm(5000, generator=torch.Generator().manual_seed(1))
    b = DeviceAveragePrecision(40, 5000, DEV)
    b.update(p[perm].to(DEV), y[perm].to(DEV))
    shuffled = b.compute()
    np.testing.assert_allclose(shuffled["ap_per_label"].cpu().numpy(), base["ap_per_label"].cpu().numpy(), rtol=1e-12)
    np.testing.assert_allclose(shuffled["map_micro"], base["map_micro"], rtol=1e-12)
    c = DeviceAveragePrecision(40, 5000, DEV)
    c.update((p.double() * 3 - 7).float().to(DEV), y.to(DEV))   # strictly increasing on the f32 grid used here? affine
    mono = c.compute()
    # affine maps can merge neighbouring f32 values; compare only up to such merges
    np.testing.assert_allclose(mono["ap_per_label"].cpu().numpy(), base["ap_per_label"].cpu().numpy(), rtol=1e-4)
    z = torch.tensor([[0.0], [-0.0], [1.0], [-1.0]])
    yz = torch.tensor([[1], [0], [0], [1]])
    d = DeviceAveragePrecision(1, 4, DEV)
    d.update(z.to(DEV), yz.to(DEV))
    np.testing.assert_allclose(d.compute()["ap_per_label"].cpu().numpy()[0],
                               MO.average_precision(z[:, 0].numpy(), yz[:, 0].numpy()), rtol=1e-12)


@pytest.mark.parametrize("n,nl,batch,T", [(50, 3, 50, 50), (3000, 130, 1024, 50), (2000, 64, 2000, 7)])
def test_binned_auprc_matches_oracle(n, nl, batch, T):
    from protnote_amd.utils.evaluation import DeviceBinnedAUPRC

    p, y = _case(n, nl, seed=3 * n + nl, saturate=True)
    p[0, 0], p[1, 0] = 0.0, 1.0                        # exactly on the end thresholds
    thr = torch.linspace(0, 1.0, T)
    p[2, 0] = thr[T // 2]                              # exactly on an inner threshold (>= counts it)
    y[:, 1] = False
    acc = DeviceBinnedAUPRC(nl, DEV, threshold=T)
    for i in range(0, n, batch):
        acc.update(p[i:i + batch].to(DEV), y[i:i + batch].to(DEV).long())
    out = acc.compute()
    # integer state: histogram of bin(p) = #{k: p >= thr_k}
    bins = (p.numpy()[:, :, None] >= thr.numpy()[None, None, :]).sum(-1)
    want_all = np.stack([np.bincount(bins[:, j], minlength=T + 1) for j in range(nl)])
    want_pos = np.stack([np.bincount(bins[:, j][y.numpy()[:, j]], minlength=T + 1) for j in range(nl)])
    np.testing.assert_array_equal(acc.all[:nl].cpu().numpy(), want_all)
    np.testing.assert_array_equal(acc.pos[:nl].cpu().numpy(), want_pos)
    np.testing.assert_array_equal(acc.all[nl].cpu().numpy(), want_all.sum(0))
    np.testing.assert_array_equal(acc.pos[nl].cpu().numpy(), want_pos.sum(0))
    per = np.array([MO.binned_auprc(p[:, j].numpy(), y[:, j].numpy(), thr.numpy()) for j in range(nl)])
    got = out["ap_per_label"].cpu().numpy()
    assert np.array_equal(np.isnan(got), np.isnan(per)) and np.isnan(got[1])
    np.testing.assert_allclose(got[~np.isnan(per)], per[~np.isnan(per)], rtol=1e-12)
    np.testing.assert_allclose(out["map_micro"], MO.binned_auprc(p.numpy().ravel(), y.numpy().ravel(), thr.numpy()),
                               rtol=1e-12)
    np.testing.assert_allclose(out["map_macro"], MO.macro_mean(per), rtol=1e-12)


def test_exact_ap_full_label_set_known_answer():
    """BASELINE-sized label set (32 102 labels) x 20 000 proteins = 642 M resident pairs, with a closed-form answer:
    label j's positives sit at every m_j-th rank of its column (ranks m, 2m, ...), so AP_j = 1/m_j exactly; all labels
    share the score-by-rank values, so the pooled ranking has n tie groups of N_L pairs each and micro AP follows from
    the per-rank positive counts.  Protein order is shuffled per batch."""
    from protnote_amd.utils.evaluation import DeviceAveragePrecision

    n, NL, batch = 20000, 32102, 2000
    g = torch.Generator().manual_seed(0)
    m = torch.randint(1, 40, (NL,), generator=g)
    m[7] = n + 1                                                    # no positive at all -> NaN
    rank_of_row = torch.randperm(n, generator=g) + 1                # protein i has rank rank_of_row[i] in every column
    acc = DeviceAveragePrecision(NL, n, DEV)
    m_d = m.to(DEV)
    for i in range(0, n, batch):
        r = rank_of_row[i:i + batch].to(DEV)
        scores = (1.0 - r.float() / n)[:, None].expand(-1, NL)      # descending in rank; f32-distinct for n = 20 000
        y = (r[:, None] % m_d[None, :]) == 0
        acc.update(scores, y)
    out = acc.compute()
    got = out["ap_per_label"].cpu().numpy()
    want = 1.0 / m.numpy().astype(np.float64)
    want[7] = np.nan
    small = (n // m.numpy()) == 0
    assert np.isnan(got[7]) and not small[np.arange(NL) != 7].any()
    keep = np.arange(NL) != 7
    np.testing.assert_allclose(got[keep], want[keep], rtol=1e-12)
    ranks = np.arange(1, n + 1)
    pos_at_rank = np.zeros(n, dtype=np.int64)
    for mj, c in zip(*np.unique(m.numpy()[keep], return_counts=True)):
        pos_at_rank[mj - 1::mj] += c
    tp = np.cumsum(pos_at_rank).astype(np.float64)
    micro = float(np.sum(pos_at_rank * tp / (ranks * NL)) / tp[-1])
    np.testing.assert_allclose(out["map_micro"], micro, rtol=1e-12)
    np.testing.assert_array_equal(out["positives_per_label"].cpu().numpy()[keep], (n // m.numpy())[keep])


@pytest.mark.parametrize("n,nl,cuts,kw", [(5000, 7, 3, dict(ties=True)), (40000, 5, 8, {}), (300, 3, 8, dict(saturate=True))])
def test_ap_partial_key_ranges_sum_to_the_whole_ranking(n, nl, cuts, kw):
    """pn_ap_partial (the per-GPU piece of the multi-GPU micro AP): cut the key space into `cuts` ranges (equal keys
    share a range, some ranges may be empty), rank every range on its own with the positives / pairs of the higher
    ranges as offsets - the partial sums add up to the AP of the whole ranking (oracle, 1e-12)."""
    from protnote_amd import _lib as L
    from protnote_amd.utils.evaluation import DeviceAveragePrecision

    p, y = _case(n, nl, seed=n + cuts, **kw)
    acc = DeviceAveragePrecision(nl, n, DEV)
    acc.update(p.to(DEV), y.to(DEV))
    keys, hits = acc.keys[:, :n].reshape(-1), acc.hits[:, :n].reshape(-1)
    k64 = keys.long() & 0xFFFFFFFF
    edges = torch.quantile(k64.double().cpu(), torch.linspace(0, 1, cuts + 1, dtype=torch.float64)[1:-1]).long().to(DEV)
    dest = (cuts - 1) - torch.bucketize(k64, edges, right=True)  # range 0 = highest keys
    total, npos_total, tp_before, k_before = 0.0, 0, 0, 0
    for v in range(cuts):
        sel = dest == v
        mk, mh = keys[sel].contiguous(), hits[sel].contiguous()
        m = int(mk.numel())
        part = torch.zeros(1, dtype=torch.float64, device=DEV)
        npos = torch.zeros(1, dtype=torch.int64, device=DEV)
        nbytes = L.lib().pn_ap_partial_ws_bytes(m)
        ws = torch.empty(nbytes, dtype=torch.uint8, device=DEV)
        L.check(L.lib().pn_ap_partial(L.ptr(mk) if m else None, L.ptr(mh) if m else None, m, tp_before, k_before,
                                      L.ptr(part), L.ptr(npos), L.ptr(ws), nbytes, L.stream_ptr()))
        torch.cuda.synchronize()
        assert int(npos) == int(mh.sum())
        total += float(part)
        npos_total += int(npos)
        tp_before += int(npos)
        k_before += m
    micro = MO.average_precision_fast(p.numpy().ravel(), y.numpy().ravel())
    np.testing.assert_allclose(total / npos_total, micro, rtol=1e-12)
