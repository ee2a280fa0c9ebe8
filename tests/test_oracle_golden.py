"""Pin the CPU oracle (oracle/qpal_oracle.c) to the golden vectors generated from the reference's own
Python (tests/golden/make_golden.py).  Bit-exact everywhere: integer states/indices and fp16 weights."""
import os

import numpy as np
import pytest

from oracle import oracle

TCQ_COMBOS = [(9, kv) for kv in range(2, 11)] + [(10, 8), (10, 9), (10, 10), (11, 9), (11, 10)]


@pytest.fixture(scope="module")
def tcq(golden_dir):
    return np.load(os.path.join(golden_dir, "tcq.npz"))


@pytest.fixture(scope="module")
def lut_tc(golden_dir):
    return np.load(os.path.join(golden_dir, "lut_tc.npz"))


@pytest.fixture(scope="module")
def simt(golden_dir):
    return np.load(os.path.join(golden_dir, "simt.npz"))


@pytest.mark.parametrize("S,KV", TCQ_COMBOS)
def test_tcq_states_and_weights(tcq, S, KV):
    m, k = int(tcq["m"]), int(tcq["k"])
    tr = tcq[f"tcq_S{S}_KV{KV}_trellis"]
    assert tr.shape == (m // 16 * (k // 16), 8 * KV) and tr.dtype == np.int16
    st = oracle.tcq_states(tr, m, k, KV).reshape(-1, 128)
    assert np.array_equal(st, tcq[f"tcq_S{S}_KV{KV}_states"])
    W = oracle.tcq_dequant(tr, tcq[f"tcq_S{S}_KV{KV}_tlut"], m, k, S, KV)
    assert np.array_equal(W.view(np.uint16), tcq[f"tcq_S{S}_KV{KV}_W"].view(np.uint16))


@pytest.mark.parametrize("KV", [2, 5, 6, 9])
def test_tcq_comb_and_combt_are_concatenations(tcq, KV):
    """comb = row halves, combt = column halves (comb_linear.py:35-48, 178-191)."""
    m, k = int(tcq["m"]), int(tcq["k"])
    S = 9
    t1, t2 = tcq[f"tcq_S{S}_KV{KV}_trellis"], tcq[f"tcq_S{S}_KV{KV + 1}_trellis"]
    tl = tcq[f"tcq_S{S}_KV{KV}_tlut"]
    W1, W2 = tcq[f"tcq_S{S}_KV{KV}_W"], tcq[f"tcq_S{S}_KV{KV + 1}_W"]
    comb = oracle.tcq_dequant(t1, tl, 2 * m, k, S, KV, c2=t2, KV2=KV + 1, split=1)
    assert np.array_equal(comb.view(np.uint16), np.concatenate([W1, W2], 0).view(np.uint16))
    combt = oracle.tcq_dequant(t1, tl, m, 2 * k, S, KV, c2=t2, KV2=KV + 1, split=2)
    assert np.array_equal(combt.view(np.uint16), np.concatenate([W1, W2], 1).view(np.uint16))


@pytest.mark.parametrize("vec,bits", [(1, b) for b in range(2, 9)] + [(2, b) for b in range(2, 13)])
def test_lut_tc(lut_tc, vec, bits):
    m, k = int(lut_tc["m"]), int(lut_tc["k"])
    q = lut_tc[f"tc_v{vec}_b{bits}_qweight"]
    assert q.shape == (m, bits * k // 32 // vec)
    idx = oracle.lut_tc_indices(q, m, k, bits, vec)
    assert np.array_equal(idx, lut_tc[f"tc_v{vec}_b{bits}_idx"])
    W = oracle.lut_tc_dequant(q, lut_tc[f"tc_v{vec}_b{bits}_lut"], m, k, bits, vec)
    assert np.array_equal(W.view(np.uint16), lut_tc[f"tc_v{vec}_b{bits}_W"].view(np.uint16))


@pytest.mark.parametrize("vec,bits", [(1, b) for b in range(2, 9)] + [(2, b) for b in range(3, 13)]
                         + [(4, b) for b in range(6, 13)])
def test_simt_indices(simt, vec, bits):
    m, k = int(simt["m"]), int(simt[f"simt_v{vec}_k"])
    q = simt[f"simt_v{vec}_b{bits}_qweight"]
    idx = oracle.simt_indices(q, m, k, bits, vec)
    assert np.array_equal(idx, simt[f"simt_v{vec}_b{bits}_idx"])


@pytest.mark.parametrize("vec,bits", [(1, 3), (1, 4), (1, 8), (2, 5), (2, 8), (2, 12)])
def test_tc_and_simt_views_of_one_matrix_agree(simt, vec, bits):
    m, k = int(simt["conv_m"]), int(simt["conv_k"])
    tc, sm = simt[f"conv_v{vec}_b{bits}_tc"], simt[f"conv_v{vec}_b{bits}_simt"]
    want = simt[f"conv_v{vec}_b{bits}_idx"]
    assert np.array_equal(oracle.lut_tc_indices(tc, m, k, bits, vec), want)
    assert np.array_equal(oracle.simt_indices(sm, m, k, bits, vec), want)


def test_gemv_matches_numpy(tcq):
    W = tcq["tcq_S9_KV6_W"]
    rng = np.random.default_rng(0)
    x = rng.standard_normal((3, W.shape[1])).astype(np.float16)
    out, aout = oracle.gemv(W, x)
    ref = x.astype(np.float64) @ W.astype(np.float64).T
    assert np.allclose(out, ref, rtol=0, atol=1e-12 * aout.max())
    cpu = oracle.cpu_tcq_linear(tcq["tcq_S9_KV6_trellis"], None, tcq["tcq_S9_KV6_tlut"], x,
                                W.shape[0], 3, W.shape[1], 9, 6, 0, 0)
    assert np.all(np.abs(cpu - ref) <= 1e-5 * aout + 1e-30)
