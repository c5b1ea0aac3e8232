"""The --loop-* front end: flag names, defaults and setLoopParam's checks as in
tests/test_params_mugiq.cpp:12-24,77-112 and tests/loop.cpp:620-748, then the whole driver flow on synthetic inputs."""
import numpy as np
import pytest

from util import orc, momenta_p2_le, rel_err


def _args(cli, argv):
    return cli.build_parser().parse_args(argv)


def test_flag_names_defaults_and_checks(hip, tmp_path):
    from mugiq_amd import loop_cli as cli
    mom = tmp_path / "momenta.txt"
    mom.write_text("0 0 0\n1 0 0\n-1 2 0\n")
    a = _args(cli, [])
    assert (a.mugiq_mom_filename, a.loop_gauge_filename, a.loop_ft_sign, a.loop_calc_type) == ("momenta.txt", "", None, None)
    assert (a.loop_write_mom_space_hdf5, a.loop_write_pos_space_hdf5, a.loop_doMomProj, a.loop_doNonLocal) == (True, False, True, True)
    with pytest.raises(cli.LoopParamError, match="Loop FT sign is undefined"):
        cli.setLoopParam(a)
    a = _args(cli, ["--loop-ft-sign", "minus"])
    with pytest.raises(cli.LoopParamError, match="Loop Calculation Type is undefined"):
        cli.setLoopParam(a)
    base = ["--loop-ft-sign", "minus", "--loop-calc-type", "opt", "--momenta-filename", str(mom)]
    with pytest.raises(cli.LoopParamError, match="--loop-mom-space-filename"):
        cli.setLoopParam(_args(cli, base))
    with pytest.raises(cli.LoopParamError, match="--displace-entry-string is not set"):
        cli.setLoopParam(_args(cli, base + ["--loop-mom-space-filename", "x.h5"]))
    with pytest.raises(cli.LoopParamError, match="Cannot open file"):
        cli.setLoopParam(_args(cli, ["--loop-ft-sign", "plus", "--loop-calc-type", "basic", "--loop-write-mom-space", "no",
                                     "--loop-do-nonlocal", "no", "--momenta-filename", str(tmp_path / "none.txt")]))
    with pytest.raises(SystemExit):
        _args(cli, ["--loop-ft-sign", "up"])
    p = cli.setLoopParam(_args(cli, base + ["--loop-mom-space-filename", "x.h5", "--displace-entry-string", "+z:1,8;-x:3;+y:5,2"]))
    assert (p.FTSign, p.calcType, p.doMomProj, p.doNonLocal, p.writeMomSpaceHDF5, p.writePosSpaceHDF5) == \
        (-1, hip.LOOP_CALC_TYPE_OPT_KERNEL, True, True, True, False)
    assert p.disp_entry == ["+z:1,8", "-x:3", "+y:5,2"] and p.disp_str == ["+z", "-x", "+y"]
    assert p.disp_start == [1, 3, 5] and p.disp_stop == [8, 3, 2]              # the engine swaps start > stop with a warning
    assert p.Nmom == 3 and p.momMatrix == [[0, 0, 0], [1, 0, 0], [-1, 2, 0]] and p.fname_mom_h5 == "x.h5"
    p = cli.setLoopParam(_args(cli, ["--loop-ft-sign", "plus", "--loop-calc-type", "basic", "--loop-write-mom-space", "no",
                                     "--loop-do-nonlocal", "no", "--loop-do-momproj", "no", "--momenta-filename", str(mom)]))
    assert (p.FTSign, p.calcType, p.doMomProj, p.doNonLocal, p.disp_str) == (1, hip.LOOP_CALC_TYPE_BASIC_KERNEL, False, False, [])
    (tmp_path / "bad.txt").write_text("0 0 0\n1 1\n")
    with pytest.raises(cli.LoopParamError, match="Incorrect file format in Line 1"):
        cli.setLoopParam(_args(cli, ["--loop-ft-sign", "plus", "--loop-calc-type", "opt", "--loop-write-mom-space", "no",
                                     "--loop-do-nonlocal", "no", "--momenta-filename", str(tmp_path / "bad.txt")]))


@pytest.mark.gpu
@pytest.mark.parametrize("prec,calc", [("double", "opt"), ("single", "basic")])
def test_command_line_run_matches_oracle(hip, tmp_path, capsys, prec, calc):
    """`tests/loop`'s flow for configs[0] (8^4, N_ev = 4) plus displacements: the HDF5 file the command line writes holds
    what the oracle computes from the same synthetic inputs."""
    import json
    import h5read
    from mugiq_amd import loop_cli as cli
    try:
        h5 = h5read.H5()
    except ImportError:
        pytest.skip("libhdf5 not available")
    moms = momenta_p2_le(2)
    mom = tmp_path / "momenta.txt"
    mom.write_text("".join("%d %d %d\n" % m for m in moms))
    out = tmp_path / "loops.h5"
    entry = "+z:1,2;-x:3;-t:2,1"
    argv = ["--dim", "8", "8", "8", "8", "--prec", prec, "--n-ev", "4", "--seed", "4321", "--loop-ft-sign", "minus", "--loop-calc-type", calc,
            "--momenta-filename", str(mom), "--displace-entry-string", entry, "--loop-mom-space-filename", str(out)]
    assert cli.main(argv) == 0
    info = json.loads(capsys.readouterr().out.strip().splitlines()[-1])
    assert info["nLoop"] == 1 + 2 + 1 + 2 and info["Nmom"] == len(moms) and info["data"].startswith("synthetic")
    # the same synthetic inputs, through the oracle
    args = cli.build_parser().parse_args(argv)
    fields, sigma, gauge = cli.synthetic_inputs(args)
    X = (8, 8, 8, 8)
    ev = [f.get_logical().astype(np.complex128) for f in fields]
    Uo = gauge.get_logical().astype(np.complex128)
    _, s, a, b = orc.parse_disp_entry_string(entry)
    cprm = orc.LoopComputeParam(s, a, b)
    sg = np.float32(sigma).astype(np.float64) if prec == "single" else sigma
    pos = orc.compute_loop_position_space(ev, sg, cprm, Uo, X)
    V, locV3 = 8 ** 4, 8 ** 3
    ref = orc.momentum_projection_local(orc.convert_idx_order_map_gamma(pos, cprm.nData, cprm.nLoop, 2, V // 2, X),
                                        orc.phase_matrix(moms, locV3, -1, X, X), 8, cprm.nData, locV3, len(moms))
    ref = np.asarray(ref).reshape(len(moms), cprm.nLoop, 16, 8)
    names = ["disp_0", "disp_+z_1", "disp_+z_2", "disp_-x_3", "disp_-t_1", "disp_-t_2"]
    fid = h5.open(str(out))
    scale = np.abs(ref).max()
    tol = 1e-12 if prec == "double" else 1e-5
    for im, p in enumerate(moms):
        for iL, dn in enumerate(names):
            for ig in range(16):
                got = h5.read(fid, "/mom_%+d_%+d_%+d/%s/%s/loop" % (tuple(p) + (dn, hip.GammaName(ig))))
                assert np.abs(got[:, 0] + 1j * got[:, 1] - ref[im, iL, ig]).max() < tol * scale, (p, dn, ig)
    h5.close(fid)
