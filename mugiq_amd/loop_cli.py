"""The loop options of the reference's drivers, under the same flag names (tests/test_params_mugiq.cpp:77-112), and the
`setLoopParam` filler of tests/loop.cpp:620-748 -- so a command line written for `tests/loop` carries over unchanged.

The reference's driver then runs QUDA's eigensolver; that part is out of scope here (eigenvectors and sigma are the
INPUTS of the loop engine), so `main()` feeds the engine synthetic eigenvectors of the requested shape:

    python -m mugiq_amd.loop_cli --dim 8 8 8 8 --prec double --n-ev 4 \\
        --loop-ft-sign minus --loop-calc-type opt --momenta-filename momenta.txt \\
        --displace-entry-string "+z:1,2;-x:3" --loop-mom-space-filename loops.h5

Multi-GPU: launch with torchrun and pass --gridsize gx gy gz gt (QUDA's --gridsize); one process per GPU.
"""
import argparse
import json
import os
import sys

import numpy as np

from .loop import (LOOP_CALC_TYPE_BASIC_KERNEL, LOOP_CALC_TYPE_BLAS, LOOP_CALC_TYPE_OPT_KERNEL, MugiqLoopParam, read_momenta_file,
                   parseDisplaceEntryString)

LOOP_FT_SIGN = {"plus": 1, "minus": -1}                                                     # loop_ft_sign_map
LOOP_CALC_TYPE = {"blas": LOOP_CALC_TYPE_BLAS, "opt": LOOP_CALC_TYPE_OPT_KERNEL, "basic": LOOP_CALC_TYPE_BASIC_KERNEL}
YES_NO = {"yes": True, "no": False}


def _choice(mapping):
    def conv(s):
        if s not in mapping:
            raise argparse.ArgumentTypeError("options are %s" % "/".join(mapping))
        return mapping[s]
    return conv


def add_loop_option_mugiq(parser):
    """add_loop_option_mugiq (tests/test_params_mugiq.cpp:77-112): same names, same defaults (tests/test_params_mugiq.cpp:12-24)."""
    g = parser.add_argument_group("Loop-MuGiq", "Loop Options within MuGiq")
    g.add_argument("--momenta-filename", dest="mugiq_mom_filename", default="momenta.txt",
                   help="Filename with the momenta for Fourier Transform of the loop (default 'momenta.txt')")
    g.add_argument("--loop-gauge-filename", dest="loop_gauge_filename", default="",
                   help="Gauge field that will be used for non-local currents (default ''); here: a .npy file holding the "
                        "local QDP-order links [4][V*18] (tests/loop.cpp:88,106), or empty for a synthetic random SU(3) field")
    g.add_argument("--loop-ft-sign", dest="loop_ft_sign", type=_choice(LOOP_FT_SIGN), default=None,
                   help="Sign of the Loop Fourier Transform phase (default NULL, options are plus/minus)")
    g.add_argument("--loop-calc-type", dest="loop_calc_type", type=_choice(LOOP_CALC_TYPE), default=None,
                   help="Type of loop calculation (default NULL, options are blas/opt/basic)")
    g.add_argument("--loop-write-mom-space", dest="loop_write_mom_space_hdf5", type=_choice(YES_NO), default=True,
                   help="Whether to write momentum-space loop data in HDF5 format (default yes, options are yes/no)")
    g.add_argument("--loop-write-pos-space", dest="loop_write_pos_space_hdf5", type=_choice(YES_NO), default=False,
                   help="Whether to write position-space loop data in HDF5 format (default no, options are yes/no)")
    g.add_argument("--loop-do-momproj", dest="loop_doMomProj", type=_choice(YES_NO), default=True,
                   help="Whether to perform momentum projection (Fourier Transform) on the disconnected quark loop (default yes)")
    g.add_argument("--loop-do-nonlocal", dest="loop_doNonLocal", type=_choice(YES_NO), default=True,
                   help="Whether to compute quark loops for non-local currents, requires --displace-entry-string (default yes)")
    g.add_argument("--displace-entry-string", dest="disp_entry_string", default="",
                   help="Set displacement entries in the form, e.g: +z:1,8;-x:3;+y:2,5.")
    g.add_argument("--loop-mom-space-filename", dest="fname_mom_h5", default="",
                   help="Complete path to the HDF5 filename for the momentum-space loop data")
    g.add_argument("--loop-pos-space-filename", dest="fname_pos_h5", default="",
                   help="Complete path to the HDF5 filename for the position-space loop data")
    return g


class LoopParamError(ValueError):
    """what the reference reports through errorQuda in setLoopParam"""


def setLoopParam(args, gauge=None):
    """setLoopParam (tests/loop.cpp:620-748): checks, messages and field assignments in the reference's order."""
    if args.loop_ft_sign is None:
        raise LoopParamError("setLoopParam: Loop FT sign is undefined/unsupported. Options are --loop-ft-sign plus/minus")
    if args.loop_calc_type is None:
        raise LoopParamError("setLoopParam: Loop Calculation Type is undefined/unsupported. Options are --loop-calc-type blas/opt/basic")
    p = MugiqLoopParam()
    p.FTSign = args.loop_ft_sign
    p.calcType = args.loop_calc_type
    p.writeMomSpaceHDF5 = args.loop_write_mom_space_hdf5
    p.writePosSpaceHDF5 = args.loop_write_pos_space_hdf5
    p.doMomProj = args.loop_doMomProj
    p.doNonLocal = args.loop_doNonLocal
    if args.loop_write_mom_space_hdf5 and len(args.fname_mom_h5) == 0:
        raise LoopParamError("Got --loop-write-mom-space yes but no filename was given. Set option --loop-mom-space-filename")
    if args.loop_write_pos_space_hdf5 and len(args.fname_pos_h5) == 0:
        raise LoopParamError("Got --loop-write-pos-space yes but no filename was given. Set option --loop-pos-space-filename")
    p.fname_mom_h5 = args.fname_mom_h5
    p.fname_pos_h5 = args.fname_pos_h5
    if p.doNonLocal:                                                                        # tests/loop.cpp:656-705
        if len(args.disp_entry_string) == 0:
            raise LoopParamError("Got option '--loop-do-nonlocal yes' but option --displace-entry-string is not set!")
        p.disp_entry, p.disp_str, p.disp_start, p.disp_stop = parseDisplaceEntryString(args.disp_entry_string)
    p.gauge = gauge
    if not os.path.exists(args.mugiq_mom_filename):                                         # tests/loop.cpp:721-722
        raise LoopParamError("setLoopParam: Cannot open file %s to read momenta (option --momenta-filename)" % args.mugiq_mom_filename)
    try:
        p.momMatrix = read_momenta_file(args.mugiq_mom_filename)
    except ValueError as e:
        raise LoopParamError("setLoopParam: %s" % e)
    p.Nmom = len(p.momMatrix)
    return p


def build_parser():
    ap = argparse.ArgumentParser(prog="python -m mugiq_amd.loop_cli", description="disconnected quark loops from low modes (MuGiq loop driver, MI355X)")
    # the lattice options of QUDA's command line that the loop path depends on
    ap.add_argument("--dim", type=int, nargs=4, default=[8, 8, 8, 8], metavar=("X", "Y", "Z", "T"), help="LOCAL lattice dimensions (QUDA --dim)")
    ap.add_argument("--gridsize", type=int, nargs=4, default=[1, 1, 1, 1], help="process grid (QUDA --gridsize); launch with torchrun")
    ap.add_argument("--prec", choices=["double", "single"], default="double", help="eigenvector / link precision")
    ap.add_argument("--loop-prec", choices=["same", "double"], default="same", help="precision of the loop buffers and FT (double over single = mixed mode)")
    ap.add_argument("--field-order", type=int, choices=[2, 4], default=None, help="QUDA field order of the eigenvectors (default: FLOAT2 for double, FLOAT4 for single)")
    ap.add_argument("--n-ev", type=int, default=4, help="number of (synthetic) eigenvectors")
    ap.add_argument("--seed", type=int, default=777)
    add_loop_option_mugiq(ap)
    return ap


def synthetic_inputs(args, rank=0, comm=None):
    """The stand-in for the eigensolver: seeded eigenvectors (N(0,1) components, unit norm), sigma_n = 0.01 + 0.002 n,
    and -- for displaced loops -- the gauge field (--loop-gauge-filename, or random SU(3) links)."""
    import torch
    from . import GaugeField, SpinorField
    prec = 8 if args.prec == "double" else 4
    order = args.field_order or (2 if prec == 8 else 4)
    X = tuple(args.dim)
    vcb = int(np.prod(X)) // 2
    cdt = torch.complex128 if prec == 8 else torch.complex64
    gen = torch.Generator(device="cuda").manual_seed(args.seed + 1000 * rank)
    gauge = None
    if args.loop_doNonLocal and len(args.disp_entry_string) > 0:
        brd = [2 if args.gridsize[d] > 1 else 0 for d in range(4)]                         # lib/displace.cpp:16
        gauge = GaugeField(X, brd, prec)
        if args.loop_gauge_filename:
            qdp = np.load(args.loop_gauge_filename, allow_pickle=False)                    # [4][V*18] reals, QDP order
        else:
            m = torch.complex(torch.randn(4 * 2 * vcb, 3, 3, dtype=torch.float64, device="cuda", generator=gen),
                              torch.randn(4 * 2 * vcb, 3, 3, dtype=torch.float64, device="cuda", generator=gen))
            r0 = m[:, 0] / torch.linalg.vector_norm(m[:, 0], dim=-1, keepdim=True)
            r1 = m[:, 1] - (r0.conj() * m[:, 1]).sum(-1, keepdim=True) * r0
            r1 = r1 / torch.linalg.vector_norm(r1, dim=-1, keepdim=True)
            r2 = torch.linalg.cross(r0.conj(), r1.conj())                                  # det = 1
            u = torch.stack([r0, r1, r2], dim=1).reshape(4, 2 * vcb, 9)
            qdp = torch.view_as_real(u).reshape(4, 2 * vcb * 18).cpu().numpy()
        gauge.set_from_qdp_host(qdp, comm)
    fields = []
    for n in range(args.n_ev):
        f = SpinorField(X, prec, order)
        w = torch.complex(torch.randn(f.data.numel(), dtype=torch.float64, device="cuda", generator=gen),
                          torch.randn(f.data.numel(), dtype=torch.float64, device="cuda", generator=gen))
        f.data.copy_((w / torch.linalg.vector_norm(w)).to(cdt))
        fields.append(f)
    sigma = 0.01 + 0.002 * np.arange(args.n_ev)
    return fields, sigma, gauge


def main(argv=None):
    args = build_parser().parse_args(argv)
    import torch
    import torch.distributed as dist
    from . import GridComm, Loop_Mugiq

    world = int(os.environ.get("WORLD_SIZE", "1"))
    comm = None
    if world > 1:
        torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", "0")))
        dist.init_process_group(os.environ.get("MUGIQ_BACKEND", "nccl"))
        comm = GridComm(args.gridsize, device="cuda:%d" % torch.cuda.current_device())
    elif int(np.prod(args.gridsize)) != 1:
        raise SystemExit("--gridsize %s needs %d processes (torchrun)" % (args.gridsize, int(np.prod(args.gridsize))))
    rank = dist.get_rank() if world > 1 else 0
    fields, sigma, gauge = synthetic_inputs(args, rank, comm)
    prm = setLoopParam(args, gauge)
    if args.loop_prec == "double":
        prm.loopPrecision = 8

    loop = Loop_Mugiq(prm, fields, sigma, comm)
    if rank == 0:
        loop.printLoopComputeParams(lambda line: print(line, file=sys.stderr))
    loop.computeCoarseLoop()
    if prm.doMomProj and prm.writeMomSpaceHDF5:
        loop.writeLoopsHDF5()
    if rank == 0:
        print(json.dumps({"local_dim": list(args.dim), "gridsize": args.gridsize, "n_ev": args.n_ev, "prec": args.prec,
                          "field_order": fields[0].order, "nLoop": loop.nLoop, "nData": loop.nData, "Nmom": prm.Nmom,
                          "calcType": prm.calcType, "FTSign": prm.FTSign,
                          "mom_space_file": prm.fname_mom_h5 if (prm.doMomProj and prm.writeMomSpaceHDF5) else None,
                          "data": "synthetic eigenvectors (seeded N(0,1), unit norm), sigma_n = 0.01 + 0.002 n"}))
    loop.close()
    if world > 1:
        dist.destroy_process_group()
    return 0


if __name__ == "__main__":
    sys.exit(main())
