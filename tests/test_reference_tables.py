"""Pin the oracle's (and the product's) gamma tables to the literal numbers in the reference's own header.

The reference holds no golden vectors and cannot be built here, but its DeGrand-Rossi tables, gamma names and
gamma5-map are plain numeric initialisers in include/gamma.h.  Where /root/reference is mounted (the build
container; never the GPU box) this test reads that header AS TEXT, extracts the numbers and compares them with the
oracle's restatement and with what libmugiq_hip.so reports.  Nothing of the reference is stored in this repository.
"""
import os
import re

import numpy as np
import pytest

from util import orc

REF = "/root/reference/include/gamma.h"
pytestmark = pytest.mark.skipif(not os.path.exists(REF), reason="reference sources are not mounted on this machine")


def _ints(block):
    return [int(v) for v in re.findall(r"-?\d+", re.sub(r"//[^\n]*", "", block))]


def _initialiser(src, name):
    """the brace-balanced initialiser that follows `name[...][...] =`"""
    i = src.index(name)
    i = src.index("=", i)
    j = src.index("{", i)
    depth, k = 0, j
    while True:
        depth += {"{": 1, "}": -1}.get(src[k], 0)
        k += 1
        if depth == 0:
            break
    return src[j:k]


@pytest.fixture(scope="module")
def header():
    return open(REF).read()


def test_row_values_and_column_indices_equal_the_reference_header(header, hip):
    rv = np.array(_ints(_initialiser(header, "rowValue["))).reshape(16, 4, 2)          # include/gamma.h:32-49
    ci = np.array(_ints(_initialiser(header, "columnIdx["))).reshape(16, 4)            # include/gamma.h:53-71
    ref_rv = rv[..., 0] + 1j * rv[..., 1]
    assert np.array_equal(orc.GAMMA_ROW_VALUE, ref_rv)
    assert np.array_equal(orc.GAMMA_COLUMN_INDEX, ci)
    p_rv, p_ci, p_sign, p_idx = hip.gammaTables()
    assert np.array_equal(p_rv, ref_rv) and np.array_equal(p_ci, ci)


def test_gamma_names_equal_the_reference_header(header, hip):
    block = header[header.index("gNames"):]
    names = re.findall(r'"([^"]*)"', block[:block.index("}")])                          # include/gamma.h:13-18
    names = [n.strip() for n in names]
    assert len(names) == 16
    assert orc.GAMMA_NAMES == names
    assert [hip.GammaName(m) for m in range(16)] == names


def test_gamma5_map_equals_the_reference_header(header, hip):
    m = re.search(r"minusG\s*\{([^}]*)\}", header)                                       # include/gamma.h:99-102
    assert m is not None
    minus = _ints(m.group(1))
    assert sorted(minus) == [3, 6, 9, 11, 12, 14]
    sign = np.ones(16)
    sign[minus] = -1.0                                                                   # lib/contract_wrappers.cu:31-33
    assert np.array_equal(orc.gamma_map_sign(), sign)
    assert re.search(r"idxG\.at\(i\)\s*=\s*N_GAMMA_\s*-\s*i\s*-\s*1", header)            # include/gamma.h:105-109
    assert orc.INDEX_MAP_GAMMA == [15 - i for i in range(16)]
    p_rv, p_ci, p_sign, p_idx = hip.gammaTables()
    assert np.array_equal(np.asarray(p_sign, dtype=float), sign) and list(p_idx) == [15 - i for i in range(16)]


def test_displacement_flag_table_equals_the_reference_header(hip):
    src = open("/root/reference/include/displace.h").read()
    m = re.search(r"DisplaceFlagArray\s*\{([^}]*)\}", src)                               # include/displace.h:21
    flags = re.findall(r'"([^"]*)"', m.group(1))
    assert flags == list(orc.DISPLACE_FLAGS)
    for i, f in enumerate(flags):                                                        # flag -> dir = flag / 2, even flag = '+'
        assert orc.parse_displacement(f) == (i // 2, orc.DISP_SIGN_PLUS if i % 2 == 0 else orc.DISP_SIGN_MINUS)
        assert hip.parseDisplacement(f) == orc.parse_displacement(f)


def test_hdf5_group_formats_equal_the_reference_source():
    """The on-disk tree is built from three snprintf formats and two buffer sizes (lib/loop_mugiq.cpp:581-610); the
    writer must use the same ones."""
    ref = open("/root/reference/lib/loop_mugiq.cpp").read()
    ours = open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "mugiq_amd", "csrc", "hdf5_writer.cpp")).read()
    fmts = re.findall(r'snprintf\(\s*group\d_tag\s*,\s*sizeof\(group\d_tag\)\s*,\s*"([^"]*)"', ref)
    assert fmts[:3] == ["mom_%+d_%+d_%+d", "disp_0", "disp_%s_%d"]
    for f in fmts[:3]:
        assert '"%s"' % f in ours, f
    sizes = dict(re.findall(r"char (group[12]_tag)\[(\d+)\]", ref))
    assert sizes == {"group1_tag": "16", "group2_tag": "10"}
    assert dict(re.findall(r"char (group[12]_tag)\[(\d+)\]", ours)) == sizes


def _c_enum(src, name):
    """name -> {enumerator: value} of a C enum in `src` (implicit values count up from the previous one)"""
    body = re.search(r"typedef enum %s\s*\{(.*?)\}" % name, re.sub(r"//[^\n]*", "", src), re.S).group(1)
    out, nxt = {}, 0
    for item in [x.strip() for x in body.split(",") if x.strip()]:
        if "=" in item:
            k, v = [t.strip() for t in item.split("=")]
            if not re.fullmatch(r"-?\d+", v):
                continue                                                   # "= MUGIQ_INVALID_ENUM"
            nxt = int(v)
        else:
            k = item
        out[k] = nxt
        nxt += 1
    return out


def test_enumerator_values_equal_the_reference_header(hip):
    src = open("/root/reference/include/enum_mugiq.h").read()              # include/enum_mugiq.h:29-85
    inc = open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "include", "mugiq_hip.h")).read()
    mine = {k: int(v) for k, v in re.findall(r"#define (MUGIQ_HIP_[A-Z_]+) (-?\d+)\b", inc)}
    calc = _c_enum(src, "LoopCalcType_s")
    assert {k: mine["MUGIQ_HIP_" + k] for k in ("LOOP_CALC_TYPE_BLAS", "LOOP_CALC_TYPE_OPT_KERNEL", "LOOP_CALC_TYPE_BASIC_KERNEL")} == \
        {k: calc[k] for k in ("LOOP_CALC_TYPE_BLAS", "LOOP_CALC_TYPE_OPT_KERNEL", "LOOP_CALC_TYPE_BASIC_KERNEL")}
    assert (hip.LOOP_CALC_TYPE_BLAS, hip.LOOP_CALC_TYPE_OPT_KERNEL, hip.LOOP_CALC_TYPE_BASIC_KERNEL) == \
        (calc["LOOP_CALC_TYPE_BLAS"], calc["LOOP_CALC_TYPE_OPT_KERNEL"], calc["LOOP_CALC_TYPE_BASIC_KERNEL"])
    d = _c_enum(src, "DisplaceDir_s")
    assert [mine["MUGIQ_HIP_DISP_DIR_" + c] for c in "XYZT"] == [d["DispDir_" + c] for c in "xyzt"]
    sg = _c_enum(src, "DisplaceSign_s")
    assert (mine["MUGIQ_HIP_DISP_SIGN_MINUS"], mine["MUGIQ_HIP_DISP_SIGN_PLUS"]) == (sg["DispSignMinus"], sg["DispSignPlus"])
    assert (hip.DispSignMinus, hip.DispSignPlus) == (sg["DispSignMinus"], sg["DispSignPlus"])
    assert (orc.DISP_SIGN_MINUS, orc.DISP_SIGN_PLUS) == (sg["DispSignMinus"], sg["DispSignPlus"])
    fl = _c_enum(src, "DisplaceFlag_s")
    assert [fl["DispFlag_" + c] for c in "XxYyZzTt"] == list(range(8))     # the order DISPLACE_FLAGS relies on
    ft = _c_enum(src, "LoopFTSign_s")
    assert (ft["LOOP_FT_SIGN_MINUS"], ft["LOOP_FT_SIGN_PLUS"]) == (-1, 1)
