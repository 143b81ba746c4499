"""Pins the ResMLP oracle (and the product's BN folding) against vectors captured from the REAL
reference (tests/golden/make_golden.py).  CPU only."""
import os

import numpy as np
import pytest

from oracle import resmlp_oracle
from wtracker_amd import resmlp

# reference batch-size variance is ~1e-5; stated tolerance for the path:
ATOL, RTOL = 2e-4, 1e-5
KNOWN_F0 = {"100ms": [3.7556774616241455, 7.793441295623779], "200ms": [12.784984588623047, 15.542585372924805]}
KNOWN_SHA = {"100ms": "03416bb2566ec039", "200ms": "3dc207ce3d5f3101"}


@pytest.mark.parametrize("tag", ["100ms", "200ms"])
def test_fixture_matches_survey_known_answers(golden_dir, tag):
    z = np.load(os.path.join(golden_dir, f"resmlp_{tag}.npz"))
    np.testing.assert_allclose(z["y_zero"][0], KNOWN_F0[tag], rtol=0, atol=1e-6)
    assert bytes(z["sha256"]).decode()[:16] == KNOWN_SHA[tag]


@pytest.mark.parametrize("tag", ["100ms", "200ms"])
def test_oracle_matches_reference_outputs(golden_dir, tag):
    path = os.path.join(golden_dir, f"resmlp_{tag}.npz")
    z = np.load(path)
    st = resmlp_oracle.load_state(path)
    y = resmlp_oracle.forward(st, z["x"])
    np.testing.assert_allclose(y, z["y_batch"], rtol=RTOL, atol=ATOL)
    np.testing.assert_allclose(y, z["y_single"], rtol=RTOL, atol=ATOL)
    np.testing.assert_allclose(resmlp_oracle.forward(st, np.zeros((1, z["x"].shape[1]), np.float32)), z["y_zero"], rtol=RTOL, atol=ATOL)


@pytest.mark.parametrize("tag,shape", [("100ms", (28, 40, [10, 4, 10, 40], 4)), ("200ms", (28, 60, [20, 8, 20, 60], 6))])
def test_folded_model_structure_and_values(golden_dir, tag, shape):
    path = os.path.join(golden_dir, f"resmlp_{tag}.npz")
    z = np.load(path)
    m = resmlp.load_npz(path)
    in_dim, hidden, block_dims, n_blocks = shape
    assert m.in_dim == in_dim and m.out_dim == 2 and m.n_blocks == n_blocks and m.layers_per_block == 4
    assert [w.shape[0] for w, _, _ in m.layers[1:5]] == block_dims and m.layers[0][0].shape == (hidden, in_dim)
    assert all(r for _, _, r in m.layers[:-1]) and not m.layers[-1][2]
    assert m.macs_per_sample == {"100ms": 4720, "200ms": 18120}[tag]
    # folded affine chain in numpy == reference outputs
    x = z["x"]
    h = np.maximum(x @ m.layers[0][0].T + m.layers[0][1], 0)
    li = 1
    for _ in range(m.n_blocks):
        t = h
        for _ in range(m.layers_per_block):
            w, b, r = m.layers[li]
            t = np.maximum(t @ w.T + b, 0)
            li += 1
        h = h + t
    y = h @ m.layers[-1][0].T + m.layers[-1][1]
    np.testing.assert_allclose(y, z["y_batch"], rtol=RTOL, atol=ATOL)


@pytest.mark.parametrize("tag", ["100ms", "200ms"])
def test_plain_c_oracle_matches_reference_outputs(golden_dir, tag):
    import subprocess

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    subprocess.run(["make", "-s", "-C", os.path.join(root, "oracle")], check=True)
    path = os.path.join(golden_dir, f"resmlp_{tag}.npz")
    z = np.load(path)
    y = resmlp_oracle.c_forward(resmlp_oracle.load_state(path), z["x"])
    np.testing.assert_allclose(y, z["y_batch"], rtol=RTOL, atol=ATOL)
