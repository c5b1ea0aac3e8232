#!/usr/bin/env python3
"""Per-kernel HBM traffic table of a refresh run: PMC bytes (profiles/traffic_extra_latest.json, written by tools/pmc_traffic_extra.py)
next to the ALGORITHMIC bytes of the bench workloads (32^4 x 200 headline; 48.48.24.24 x N_ev displaced leg; 32^4 MG leg, n_vec 24,
N_ev 200).  usage: pmc_table.py <traffic_extra_latest.json> > table.txt"""
import json
import sys

t = json.load(open(sys.argv[1]))
nev = t.get("displaced_nev", 400)
V48, V32 = 48 * 48 * 24 * 24, 32 ** 4
alg = []


def a(sub, grid, gb, what):
    alg.append((sub, grid, gb, what))


a("loop_contract_kernel", V32, V32 * (200 * 192 + 256) / 1e9, "N_ev 200 x 192 B per site read once + 256 B written")
a("tile16_displaced_contract_kernel<double, double, 2, 0", None, V48 * (nev * 192 + 3 * (192 + 256)) / 1e9,
  "eigenvectors once (N_ev %d) + 3 x (W_k 192 B read + slot 256 B written) per site" % nev)
for d in (0, 1, 2, 3):
    a("mfma_tile_displaced_contract_kernel<%d," % d, None, V48 * (nev * 192 + 144 * (1 + 3 / (48 if d < 2 else 24)) + 3 * 256 + (256 if d == 1 else 0)) / 1e9,
      "eigenvectors once (N_ev %d) + the axial gauge (144 B per staged position) + 3 slots x 256 B written" % nev + (" + the carried ultra-local slot" if d == 1 else ""))
a("axial_gauge_kernel", None, V48 * (192 + 144 * (1 + 3 / 24)) / 1e9, "W_1 once (the W_k of the continued positions are a boundary term) + g written")
for d in (1, 2, 3):
    a("tile_displaced_contract_kernel<double, double, 2, %d" % d, None, V48 * (nev * 192 + 3 * (192 + 256) + (256 if d == 1 else 0)) / 1e9,
      "as above" + (" + the carried ultra-local slot (256 B written)" if d == 1 else ""))
a("eo_dft_x", None, V48 * 13 * 16 * 16 * (1 + 7 / 48) / 1e9, "13 of 25 slots (reflected ones are derived in momentum space): 16 x 13 x V complex in + 7/48 of it out")
a("coarse_outer_kernel", None, (200 * 4096 * 48 * 16 + 4096 * 48 * 48 * 16) / 1e9, "200 coarse eigenvectors (2 n_vec = 48 components, 8^4 sites) once + C(X) written")
a("fine_congruence_mfma", None, (V32 * 288 * 16 + 4096 * 48 * 48 * 16 + V32 * 256) / 1e9, "V once (12 n_vec complex per fine site) + C(X) once + 16 traces per site added")
a("coarse_pack_kernel", None, (200 * 4096 * 48 * 16 * 2) / 1e9, "coarse eigenvectors read + eigenvector-major copy written")
a("prolong_mfma_kernel", None, (V32 * 288 * 16 + V32 * 100 * 192) / 1e9, "PER PASS (two passes of 13 | 12 blocks of 8 eigenvectors): V once + ~100 fine eigenvectors written")
a("read_probe_kernel", None, 200 * V32 * 192 / 1e9, "the buffer once")
a("cov_displace_kernel", None, V48 * (192 + 144 + 192) / 1e9, "neighbour spinor + link read, spinor written (identity-spinor chain: the W_k fields)")
print("library sources %s, displaced leg at N_ev = %d" % (t["library_fingerprint"], nev))
print("%-82s %9s %3s %9s %9s %6s  %s" % ("kernel", "grid", "n", "PMC GB", "algor. GB", "ratio", "algorithmic bytes are"))
for k in t["kernels"]:
    m = None
    for sub, grid, gb, what in alg:
        if sub in k["name"] and (grid is None or grid == k["grid"]):
            m = (gb, what)
    nm = k["name"].replace("void mugiq::", "").replace("mugiq::", "")[:82]
    g = k["hbm_bytes_per_launch"] / 1e9
    if m:
        print("%-82s %9d %3d %9.2f %9.2f %6.2f  %s" % (nm, k["grid"], k["launches_averaged"], g, m[0], g / m[0], m[1]))
    else:
        print("%-82s %9d %3d %9.2f" % (nm, k["grid"], k["launches_averaged"], g))
