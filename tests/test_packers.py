"""Packers + on-disk layer format.

CPU part: the host-side encoders (C-ABI qpal_pack_*) reproduce the reference's packed bytes bit for bit on the golden
vectors (tests/golden/*.npz hold the reference's own pack_trellis / pack_qweight / pack_qweight_*_simt outputs), round-trip
through the oracle's decoders on other shapes, and refuse invalid input.  GPU part: indices -> packers -> layer file
(torch.save of the reference's `save_info` dict) -> loader -> device decode gives back lut[indices].
"""
import os

import numpy as np
import pytest
import torch

GOLD = os.path.join(os.path.dirname(__file__), "golden")
TCQ_COMBOS = [(9, kv) for kv in range(2, 11)] + [(10, 8), (10, 9), (10, 10), (11, 9), (11, 10)]


def _qp():
    import qpalette_amd
    return qpalette_amd


def _walk(rng, ntiles, KV):
    """random tail-biting trellis walks: 128 sixteen-bit windows at stride KV of a circular 128*KV-bit string"""
    bits = rng.integers(0, 2, size=(ntiles, 128 * KV), dtype=np.int64)
    ext = np.concatenate([bits, bits[:, :16]], axis=1)
    w = (1 << np.arange(15, -1, -1)).astype(np.int64)
    return np.stack([(ext[:, t * KV:t * KV + 16] * w).sum(axis=1) for t in range(128)], axis=1)


def _to_qidxs(states, m, k):
    return np.ascontiguousarray(states.reshape(m // 16, k // 16, 16, 8).transpose(0, 2, 1, 3).reshape(m, k // 2))


@pytest.mark.parametrize("S,KV", TCQ_COMBOS)
def test_pack_trellis_equals_reference(S, KV):
    qp = _qp()
    g = np.load(os.path.join(GOLD, "tcq.npz"))
    m, k = int(g["m"]), int(g["k"])
    states = g[f"tcq_S{S}_KV{KV}_states"].astype(np.int32)
    out = qp.packers.pack_trellis(torch.from_numpy(_to_qidxs(states, m, k)), m, k, KV)
    assert out.dtype == torch.int16 and tuple(out.shape) == ((m // 16) * (k // 16), 8 * KV)
    assert np.array_equal(out.numpy(), g[f"tcq_S{S}_KV{KV}_trellis"])


@pytest.mark.parametrize("vec,bits", [(1, b) for b in range(2, 9)] + [(2, b) for b in range(2, 13)])
def test_pack_qweight_equals_reference(vec, bits):
    qp = _qp()
    g = np.load(os.path.join(GOLD, "lut_tc.npz"))
    out = qp.packers.pack_qweight(torch.from_numpy(g[f"tc_v{vec}_b{bits}_idx"]), vec, bits)
    assert out.dtype == torch.int32
    assert np.array_equal(out.numpy(), g[f"tc_v{vec}_b{bits}_qweight"])


@pytest.mark.parametrize("vec,bits", [(1, b) for b in range(2, 9)] + [(2, b) for b in range(3, 13)] +
                         [(4, b) for b in range(6, 13)])
def test_pack_simt_equals_reference(vec, bits):
    qp = _qp()
    g = np.load(os.path.join(GOLD, "simt.npz"))
    idx = torch.from_numpy(g[f"simt_v{vec}_b{bits}_idx"])
    out = qp.packers.pack_qweight_sq_simt(idx, bits) if vec == 1 else qp.packers.pack_qweight_vq_simt(idx, bits, vec)
    assert np.array_equal(out.numpy().view(np.uint32), g[f"simt_v{vec}_b{bits}_qweight"].view(np.uint32))


def test_round_trips_through_the_oracle_decoders():
    from oracle import oracle
    qp = _qp()
    rng = np.random.default_rng(11)
    for m, k, KV in ((32, 32, 2), (96, 224, 7), (64, 4096, 10), (32, 14336, 5)):
        states = _walk(rng, (m // 16) * (k // 16), KV)
        packed = qp.packers.pack_trellis(torch.from_numpy(_to_qidxs(states, m, k)), m, k, KV).numpy()
        assert np.array_equal(oracle.tcq_states(packed, m, k, KV).reshape(-1, 128), states.astype(np.uint16))
    for m, k, bits, vec in ((32, 32, 3, 1), (64, 416, 8, 1), (32, 192, 11, 2), (96, 4096, 12, 2)):
        idx = rng.integers(0, 1 << bits, size=(m, k // vec), dtype=np.int64).astype(np.int32)
        packed = qp.packers.pack_qweight(torch.from_numpy(idx), vec, bits).numpy()
        assert np.array_equal(oracle.lut_tc_indices(packed, m, k, bits, vec), idx)
    for m, k, bits, vec in ((3, 32, 4, 1), (5, 1056, 7, 1), (2, 2112, 9, 2), (4, 2560, 3, 2), (3, 5248, 12, 4)):
        idx = rng.integers(0, 1 << bits, size=(m, k // vec), dtype=np.int64).astype(np.int32)   # ragged last blocks
        packed = qp.packers._pack_simt(torch.from_numpy(idx), bits, vec).numpy()
        assert np.array_equal(oracle.simt_indices(packed.view(np.uint32), m, k, bits, vec), idx)


def test_packers_refuse_invalid_input():
    qp = _qp()
    rng = np.random.default_rng(2)
    states = _walk(rng, 4, 6)
    bad = states.copy()
    bad[1, 17] ^= 0x0100                       # breaks the overlap with its neighbours: not a trellis walk
    with pytest.raises(RuntimeError):
        qp.packers.pack_trellis(torch.from_numpy(_to_qidxs(bad, 32, 32)), 32, 32, 6)
    with pytest.raises(RuntimeError):
        qp.packers.pack_trellis(torch.from_numpy(_to_qidxs(states, 32, 32)), 32, 32, 11)      # KV out of range
    with pytest.raises(RuntimeError):
        qp.packers.pack_trellis(torch.zeros(16, 16, dtype=torch.int32), 32, 32, 6)            # wrong shape
    with pytest.raises(RuntimeError):
        qp.packers.pack_qweight(torch.full((32, 32), 16, dtype=torch.int32), 1, 4)            # code >= 2^bits
    with pytest.raises(RuntimeError):
        qp.packers.pack_qweight(torch.zeros(32, 48, dtype=torch.int32), 1, 4)                 # k % 32
    with pytest.raises(RuntimeError):
        qp.packers.pack_qweight_vq_simt(torch.zeros(2, 24, dtype=torch.int32), 5, 2)          # k % (32 vec)
    scores = torch.zeros(32, 32, 16)
    scores[..., 5] = 1.0                                                                      # one-hot form -> argmax
    assert torch.equal(qp.packers.pack_qweight(scores, 1, 4),
                       qp.packers.pack_qweight(torch.full((32, 32), 5, dtype=torch.int32), 1, 4))


@pytest.mark.gpu
def test_indices_to_layer_file_to_device_decode(tmp_path):
    """quantiser output -> packers -> save_info file -> loader -> GPU decode == codebook[indices]; forward runs."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    qp = _qp()
    qp._native.lib()
    rng = np.random.default_rng(5)
    m, k = 256, 1024
    gen = torch.Generator().manual_seed(3)
    scales = {"SU": (torch.randint(0, 2, (k,), generator=gen) * 2 - 1).half(),
              "SV": (torch.randint(0, 2, (m,), generator=gen) * 2 - 1).half(),
              "Wscale": (0.01 + 0.02 * torch.rand(m, generator=gen)).half()}

    def check(qstr, linear_info, want_w, use_simt=False):
        info = dict(scales, in_features=k, out_features=m, hadU=k, hadV=m, dtype=torch.float16, scale=32.0, bias=None,
                    rot_info="all", linear_info=linear_info, quant_info=qp.mem_op.get_quant_info(qstr))
        path = tmp_path / f"{qstr}_{int(use_simt)}.pt"
        torch.save(info, path)
        layer = qp.IncoherentLinear.gen_layer_from_info(torch.load(path, weights_only=False), use_simt=use_simt).cuda()
        got = layer.linear.get_weight().cpu().numpy()
        assert np.array_equal(got.view(np.uint16), want_w.view(np.uint16)), qstr
        y = layer(torch.randn(2, k, device="cuda").half())
        assert tuple(y.shape) == (2, m) and torch.isfinite(y).all()
        if use_simt:   # (the SIMT module holds the re-packed codes; its file is not the tensor-core one)
            return
        # and back out through save_info: the file a later run loads is the same layer
        layer.save_info(tmp_path / "again.pt", quant_info=info["quant_info"])
        again = torch.load(tmp_path / "again.pt", weights_only=False)
        for key, val in linear_info.items():
            if torch.is_tensor(val):
                assert torch.equal(again["linear_info"][key], val.cpu()), key

    # TCQ: states -> trellis; W = quantlut_sym(states) in mma tile order (checked against the oracle's decode)
    from oracle import oracle
    KV, S = 7, 9
    states = _walk(rng, (m // 16) * (k // 16), KV)
    trellis = qp.packers.pack_trellis(torch.from_numpy(_to_qidxs(states, m, k)), m, k, KV)
    tlut = torch.randn(1 << S, 2, generator=gen).half()
    li = {"in_features": k, "out_features": m, "td_x": 16, "td_y": 16, "L": 16, "KV": KV, "V": 2, "tlut_bits": S,
          "dtype": torch.float16, "trellis": trellis, "tlut": tlut, "bias": None}
    check("tcq_7_0_1", li, oracle.tcq_dequant(trellis.numpy(), tlut.numpy(), m, k, S, KV))
    # VQ-2 / SQ in both packings: W = lut[idx]
    for qstr, bits, vec in (("ldlq_2_9_0_1", 9, 2), ("ldlq_1_5_0_1", 5, 1)):
        idx = rng.integers(0, 1 << bits, size=(m, k // vec), dtype=np.int64).astype(np.int32)
        lut = torch.randn(1 << bits, vec, generator=gen).half()
        want = lut.numpy()[idx].reshape(m, k)
        li = {"in_features": k, "out_features": m, "lut_bits": bits, "dtype": torch.float16, "vec_sz": vec,
              "qweight": qp.packers.pack_qweight(torch.from_numpy(idx), vec, bits), "lut": lut, "bias": None}
        check(qstr, li, want)
        check(qstr, li, want, use_simt=True)   # loader re-packs tensor-core order -> SIMT on the device
