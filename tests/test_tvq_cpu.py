"""
Oracle for the ingest / whole-tensor-quantization front end against vectors
produced by the reference's task_vectors.py / quantization_utils.py (tests/golden/tvq.npz).  CPU only.
"""
import numpy as np
import pytest

from helpers import bits_equal, load_golden
from oracle import svd_hybrid_oracle as orc


@pytest.fixture(scope="module")
def g():
    return load_golden("tvq.npz")


def test_task_vector_oracle(g):
    keys = [str(k) for k in g["keys"]]
    base = {k: g[f"base__{k}"] for k in keys}
    for t in (str(x) for x in g["tasks"]):
        ft = {k: g[f"ft__{t}__{k}"] for k in keys if f"ft__{t}__{k}" in g}
        tv = orc.task_vector(base, ft)
        ref_keys = [str(k) for k in g[f"ctv_keys__{t}"]]
        assert list(tv.keys()) == ref_keys
        for k in (str(x) for x in g[f"tv_keys__{t}"]):          # TaskVector skips int64 / uint8 entries
            assert bits_equal(tv[k], g[f"tv__{t}__{k}"]), (t, k)
    assert "b1" not in [str(k) for k in g["tv_keys__B"]] and "w2" not in [str(k) for k in g["tv_keys__C"]]


@pytest.mark.parametrize("qbit", [8, 4, 3])
def test_whole_tensor_quantizers_oracle(g, qbit):
    for k in (str(x) for x in g[f"qf__asymmetric{qbit}__keys"]):
        x = g[f"ft__A__{k}"]
        q, sc, zp = orc.asym_quantize(x, qbit)
        assert np.array_equal(q, g[f"qf__asymmetric{qbit}__q__{k}"]), k
        assert bits_equal(np.float32(sc), g[f"qf__asymmetric{qbit}__scale__{k}"])
        assert bits_equal(np.float32(zp), g[f"qf__asymmetric{qbit}__zp__{k}"])
        assert bits_equal(orc.asym_dequantize(q, sc, zp), g[f"qf__asymmetric{qbit}__deq__{k}"])
        qa, sa = orc.absmax_quantize(x, qbit)
        assert np.array_equal(qa, g[f"qf__absmax{qbit}__q__{k}"]), k
        assert bits_equal(np.float32(sa), g[f"qf__absmax{qbit}__scale__{k}"])
        assert bits_equal(orc.absmax_dequantize(qa, sa).reshape(x.shape), g[f"qf__absmax{qbit}__deq__{k}"])
