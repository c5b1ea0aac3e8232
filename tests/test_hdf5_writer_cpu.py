"""f1: the momentum-space HDF5 writer reproduces the reference's group tree (lib/loop_mugiq.cpp:529-656,
names include/gamma.h:11-20) and the hyperslab placement of every time rank.  Host-only, no GPU."""
import numpy as np
import pytest

from util import orc

h5read = pytest.importorskip("h5read")


@pytest.mark.parametrize("dtype", [np.complex128, np.complex64])
def test_hdf5_tree_and_contents(hip, tmp_path, dtype):
    try:
        h5 = h5read.H5()
    except ImportError:
        pytest.skip("libhdf5 not available")
    moms = [(0, 0, 0), (1, -2, 0), (-3, 3, 1)]
    disp_str, start, stop = ["+z", "-x"], [1, 2], [2, 2]
    nLoop = 1 + 2 + 1
    locT, totT = 4, 8
    nt = totT // locT
    rng = np.random.default_rng(4)
    data = (rng.standard_normal((nt, len(moms), nLoop, 16, locT)) + 1j * rng.standard_normal((nt, len(moms), nLoop, 16, locT))).astype(dtype)
    fn = str(tmp_path / "loop_mom.h5")
    hip.writeLoopsHDF5_Mom(fn, data.reshape(-1), moms, disp_str, start, stop, locT, totT)
    fid = h5.open(fn)
    assert sorted(h5.children(fid, "/")) == sorted(["mom_+0_+0_+0", "mom_+1_-2_+0", "mom_-3_+3_+1"])
    disp_names = ["disp_0", "disp_+z_1", "disp_+z_2", "disp_-x_2"]
    for im, p in enumerate(moms):
        g1 = "/mom_%+d_%+d_%+d" % p
        assert sorted(h5.children(fid, g1)) == sorted(disp_names)
        for iL, dn in enumerate(disp_names):
            assert sorted(h5.children(fid, g1 + "/" + dn)) == sorted(orc.GAMMA_NAMES)
            for ig, gn in enumerate(orc.GAMMA_NAMES):
                got = h5.read(fid, "%s/%s/%s/loop" % (g1, dn, gn))
                assert got.shape == (totT, 2) and got.dtype == (np.float64 if dtype == np.complex128 else np.float32)
                exp = np.concatenate([data[r, im, iL, ig] for r in range(nt)])          # time slabs in rank order
                assert np.array_equal(got[:, 0], exp.real) and np.array_equal(got[:, 1], exp.imag)
    h5.close(fid)


def test_hdf5_reference_tag_truncation_is_reported(hip, tmp_path):
    """group2_tag is char[10] in the reference (lib/loop_mugiq.cpp:600-608): "disp_+z_10" truncates to "disp_+z_1"
    and collides with length 1 -- reproduced knowingly as an error instead of a silent overwrite."""
    data = np.zeros(1 * 1 * 11 * 16 * 2, dtype=np.complex128)
    with pytest.raises(hip.MugiqHipError):
        hip.writeLoopsHDF5_Mom(str(tmp_path / "t.h5"), data, [(0, 0, 0)], ["+z"], [1], [10], 2, 2)
    with pytest.raises(hip.MugiqHipError):
        hip.writeLoopsHDF5_Mom(str(tmp_path / "nodir" / "t.h5"), data, [(0, 0, 0)], [], [], [], 2, 2)
