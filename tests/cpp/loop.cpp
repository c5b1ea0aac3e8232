// C++ driver program over include/mugiq_hip_operators.hpp: the computeLoop flow of the reference's test driver
// (tests/loop.cpp:751-936 -> lib/interface_mugiq.cpp:158-172) for BASELINE.json configs[0]: 8^4, unit gauge,
// N_ev = 4, ultra-local loop (+ optional displacements), one process.  QUDA's eigensolver is out of scope, so the
// eigenvectors are synthetic (seeded Gaussian, unit norm; sigma_n = 0.01 + 0.002 n) -- SURVEY.md section 8c.
//
// Accepts the reference's loop flags (tests/test_params_mugiq.cpp:77-112):
//   --dim x y z t  --nev N  --loop-ft-sign plus|minus  --loop-calc-type blas|opt|basic  --loop-do-momproj yes|no
//   --loop-do-nonlocal yes|no  --displace-entry-string "+z:1,8;-x:3"  --momenta-filename FILE
//   --loop-write-mom-space yes|no  --loop-mom-space-filename FILE  --check (compare with the C oracle, exit code)
//   --partition MASK   QUDA's test flag (bit d = comm_dim_partitioned_set(d)): run the partitioned code path on those axes with
//                      this one process as its own neighbour, through the library's own RCCL transport (mugiq_hip_rccl_comm_create:
//                      gauge borders and eigenvector halos as ncclSend / ncclRecv) -- a C++ host needs neither MPI nor Python for it
#include <hip/hip_runtime.h>

#include <complex>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <random>
#include <sstream>
#include <string>
#include <vector>

#include "mugiq_hip_operators.hpp"

using namespace mugiq_hip;
typedef std::complex<double> cplx;

// checker (tests only): plain-C restatement of the reference kernel, oracle/mugiq_oracle.c
extern "C" void oracle_loop_contract_f64(void *loop, const void *const *vL, const void *const *vR, const double *sigma, int nVec,
                                          long long site_begin, long long site_end, int volumeCB, long long stride,
                                          long long parity_offset, int order);

#define HIPCHK(x)                                                                      \
  do {                                                                                 \
    hipError_t e = (x);                                                                \
    if (e != hipSuccess) {                                                             \
      fprintf(stderr, "%s failed: %s\n", #x, hipGetErrorString(e));                    \
      return 2;                                                                        \
    }                                                                                  \
  } while (0)

static bool yes(const std::string &s) { return s == "yes" || s == "true" || s == "1"; }

int main(int argc, char **argv) {
  int X[4] = {8, 8, 8, 8};
  int nev = 4;
  bool check = false;
  int partitionMask = 0;
  std::string momFile;
  MugiqLoopParam lp;
  for (int i = 1; i < argc; i++) {
    std::string a = argv[i];
    auto next = [&]() -> std::string { return i + 1 < argc ? argv[++i] : ""; };
    if (a == "--dim") for (int d = 0; d < 4; d++) X[d] = atoi(next().c_str());
    else if (a == "--nev") nev = atoi(next().c_str());
    else if (a == "--loop-ft-sign") lp.FTSign = next() == "minus" ? LOOP_FT_SIGN_MINUS : LOOP_FT_SIGN_PLUS;
    else if (a == "--loop-calc-type") { std::string v = next(); lp.calcType = v == "basic" ? LOOP_CALC_TYPE_BASIC_KERNEL : v == "blas" ? LOOP_CALC_TYPE_BLAS : LOOP_CALC_TYPE_OPT_KERNEL; }
    else if (a == "--loop-do-momproj") lp.doMomProj = yes(next()) ? MUGIQ_BOOL_TRUE : MUGIQ_BOOL_FALSE;
    else if (a == "--loop-do-nonlocal") lp.doNonLocal = yes(next()) ? MUGIQ_BOOL_TRUE : MUGIQ_BOOL_FALSE;
    else if (a == "--displace-entry-string") setDisplaceEntryString(lp, next());
    else if (a == "--momenta-filename") momFile = next();
    else if (a == "--loop-write-mom-space") lp.writeMomSpaceHDF5 = yes(next()) ? MUGIQ_BOOL_TRUE : MUGIQ_BOOL_FALSE;
    else if (a == "--loop-mom-space-filename") lp.fname_mom_h5 = next();
    else if (a == "--check") check = true;
    else if (a == "--partition") partitionMask = atoi(next().c_str());
    else { fprintf(stderr, "unknown option %s\n", a.c_str()); return 2; }
  }
  if (lp.doMomProj == MUGIQ_BOOL_TRUE) {  // tests/loop.cpp:724-740: one "px py pz" triple per line
    std::ifstream f(momFile);
    if (!f) { fprintf(stderr, "Cannot open file %s to read momenta (option --momenta-filename)\n", momFile.c_str()); return 2; }
    std::string line;
    while (std::getline(f, line)) {
      std::istringstream iss(line);
      std::vector<int> m(3);
      if (iss >> m[0] >> m[1] >> m[2]) lp.momMatrix.push_back(m);
      else { fprintf(stderr, "Incorrect file format in Line %zu\n", lp.momMatrix.size()); return 2; }
    }
    lp.Nmom = (int)lp.momMatrix.size();
  }
  const int V = X[0] * X[1] * X[2] * X[3], vcb = V / 2;
  const size_t per = (size_t)24 * vcb;  // complex per field

  // synthetic eigenvectors, FLOAT2 order, pad 0
  std::mt19937_64 rng(777);
  std::normal_distribution<double> gauss;
  std::vector<std::vector<cplx>> hv(nev, std::vector<cplx>(per));
  std::vector<ColorSpinorField> eVecs(nev);
  std::vector<double> sigma(nev);
  std::vector<void *> dptr(nev);
  for (int n = 0; n < nev; n++) {
    double nrm = 0;
    for (auto &z : hv[n]) { z = cplx(gauss(rng), gauss(rng)); nrm += std::norm(z); }
    for (auto &z : hv[n]) z /= std::sqrt(nrm);
    HIPCHK(hipMalloc(&dptr[n], per * sizeof(cplx)));
    HIPCHK(hipMemcpy(dptr[n], hv[n].data(), per * sizeof(cplx), hipMemcpyHostToDevice));
    ColorSpinorField f{};
    f.data = dptr[n]; f.precision = 8; f.field_order = FLOAT2_FIELD_ORDER; f.nParity = 2; f.volumeCB = vcb; f.stride = vcb;
    f.parity_offset = (int64_t)12 * vcb;
    for (int d = 0; d < 4; d++) f.X[d] = X[d];
    eVecs[n] = f;
    sigma[n] = 0.01 + 0.002 * n;
  }
  // unit gauge (configs[0]) handed over the way tests/loop.cpp:902-918 does: host links in QDP order in loopParams.gauge[4]
  // + the gauge parameters; Loop_Mugiq builds the extended device field from them (Displace's setup path)
  std::vector<double> qdp((size_t)V * 18, 0.0);
  for (int s = 0; s < V; s++) qdp[(size_t)s * 18 + 0] = qdp[(size_t)s * 18 + 8] = qdp[(size_t)s * 18 + 16] = 1.0;
  GaugeParam gp{};
  for (int d = 0; d < 4; d++) {
    gp.X[d] = X[d];
    lp.gauge[d] = qdp.data();
  }
  gp.cpu_prec = 8;
  gp.cuda_prec = 8;
  lp.gauge_param = &gp;

  int rc = 0;
  MugiqHipRcclComm *rccl = nullptr;
  MugiqHipComm comm;
  if (partitionMask) {
    char id[128];
    const int grid[4] = {1, 1, 1, 1}, part[4] = {partitionMask & 1, (partitionMask >> 1) & 1, (partitionMask >> 2) & 1, (partitionMask >> 3) & 1};
    if (mugiq_hip_rccl_get_unique_id(id) || mugiq_hip_rccl_comm_create(&rccl, id, 0, 1, grid, part) || mugiq_hip_rccl_comm_fill(rccl, &comm)) {
      fprintf(stderr, "RCCL transport: %s\n", mugiq_hip_last_error());
      return 2;
    }
    printf("partitioned axes (x y z t): %d %d %d %d through the library's RCCL transport\n", part[0], part[1], part[2], part[3]);
  }
  try {
    Loop_Mugiq<double, FLOAT2_FIELD_ORDER> loop(&lp, eVecs, sigma, partitionMask ? &comm : nullptr);
    loop.computeCoarseLoop();
    if (lp.writeMomSpaceHDF5 == MUGIQ_BOOL_TRUE) loop.writeLoopsHDF5();
    MugiqHipLoopInfo info = loop.info();
    printf("computeLoop: lattice %d %d %d %d, N_ev = %d, nLoop = %d, Nmom = %d\n", X[0], X[1], X[2], X[3], nev, info.nLoop, info.Nmom);
    if (check) {
      // ultra-local slot vs the C oracle; Gamma = 1 slot sums to sum_n 1/sigma_n
      std::vector<cplx> ref((size_t)16 * V, 0.0);
      std::vector<const void *> ptrs(nev);
      for (int n = 0; n < nev; n++) ptrs[n] = hv[n].data();
      oracle_loop_contract_f64(ref.data(), ptrs.data(), ptrs.data(), sigma.data(), nev, 0, V, vcb, vcb, 12LL * vcb, 2);
      const cplx *pos = loop.dataPos();
      double err = 0, mx = 0, s1 = 0, expect = 0;
      for (size_t i = 0; i < ref.size(); i++) { err = std::max(err, std::abs(pos[i] - ref[i])); mx = std::max(mx, std::abs(ref[i])); }
      for (int i = 0; i < V; i++) s1 += pos[i].real();
      for (int n = 0; n < nev; n++) expect += 1.0 / sigma[n];
      printf("ultra-local max rel err vs C oracle: %.3e ; sum_x L_1(x) = %.12e (expected %.12e)\n", err / mx, s1, expect);
      if (err / mx > 1e-12 || std::abs(s1 - expect) > 1e-10 * expect) rc = 1;
      if (lp.doNonLocal == MUGIQ_BOOL_TRUE && !lp.disp_str.empty()) {
        // the loop nest of Loop_Mugiq::computeCoarseLoop (lib/loop_mugiq.cpp:455-509) call for call through the operator API:
        // Displace::setupDisplacement / doVectorDisplacement + performLoopContraction -- every slot must equal the driver's
        const size_t perLoop = (size_t)16 * V;
        cplx *nest_d = nullptr;
        HIPCHK(hipMalloc(&nest_d, perLoop * info.nLoop * sizeof(cplx)));
        HIPCHK(hipMemset(nest_d, 0, perLoop * info.nLoop * sizeof(cplx)));
        Displace<double, FLOAT2_FIELD_ORDER> displace(&lp, &eVecs[0], 8);
        ColorSpinorField fineEvecL{}, fineEvecR{};
        mugiq_hip::check(mugiq_hip_alloc_spinor_like(&fineEvecL, &eVecs[0], 0, nullptr));
        mugiq_hip::check(mugiq_hip_alloc_spinor_like(&fineEvecR, &eVecs[0], 0, nullptr));
        for (int id = -1; id < (int)lp.disp_str.size(); id++) {
          int e6[6] = {0, 0, 0, 0, 1, 0};
          if (id != -1) {
            displace.setupDisplacement(lp.disp_str[id]);
            mugiq_hip::check(mugiq_hip_loop_get_entry(loop.handle(), id, e6));
          }
          const size_t bufOffset = id == -1 ? 0 : perLoop * e6[5];
          for (int n = 0; n < nev; n++) {
            mugiq_hip::check(mugiq_hip_copy_spinor(&fineEvecL, &eVecs[n], nullptr));
            mugiq_hip::check(mugiq_hip_copy_spinor(&fineEvecR, &fineEvecL, nullptr));
            if (id == -1) {
              performLoopContraction<double, FLOAT2_FIELD_ORDER>(nest_d, &fineEvecL, &fineEvecR, sigma[n]);
              continue;
            }
            int dispCount = 0;
            for (int idisp = 1; idisp <= e6[3]; idisp++) {
              displace.doVectorDisplacement(DISPLACE_TYPE_COVARIANT, &fineEvecR, idisp);
              if (idisp >= e6[2] && idisp <= e6[3]) {
                performLoopContraction<double, FLOAT2_FIELD_ORDER>(nest_d + bufOffset + perLoop * dispCount, &fineEvecL, &fineEvecR, sigma[n]);
                dispCount++;
              }
            }
          }
        }
        std::vector<cplx> nest(perLoop * info.nLoop);
        HIPCHK(hipMemcpy(nest.data(), nest_d, nest.size() * sizeof(cplx), hipMemcpyDeviceToHost));
        double nerr = 0, nmx = 0;
        for (size_t i = 0; i < nest.size(); i++) { nerr = std::max(nerr, std::abs(nest[i] - pos[i])); nmx = std::max(nmx, std::abs(nest[i])); }
        printf("reference loop nest through Displace + performLoopContraction vs the driver, %d slots: max rel diff %.3e\n", info.nLoop, nerr / nmx);
        if (!(nerr / nmx < 1e-12)) rc = 1;
        mugiq_hip_free_spinor(&fineEvecL);
        mugiq_hip_free_spinor(&fineEvecR);
        (void)hipFree(nest_d);
      }
      if (lp.doMomProj == MUGIQ_BOOL_TRUE) {  // p = 0 of the g5 channel (output slot 15 <- T(0), sign +) = sum over space of L_1
        for (int im = 0; im < info.Nmom; im++)
          if (lp.momMatrix[im][0] == 0 && lp.momMatrix[im][1] == 0 && lp.momMatrix[im][2] == 0) {
            const cplx *mom = loop.dataMom_bcast();
            double tsum = 0;
            for (int t = 0; t < info.totT; t++) tsum += mom[t + (long long)info.locT * 15 + (long long)info.locT * 16 * info.nLoop * im].real();
            printf("p=0, g5 channel, summed over t: %.12e (expected %.12e)\n", tsum, expect);
            if (std::abs(tsum - expect) > 1e-10 * expect) rc = 1;
          }
      }
    }
  } catch (const Error &e) {
    fprintf(stderr, "mugiq_hip error %d: %s\n", e.status, e.what());
    rc = 1;
  }
  for (void *p : dptr) (void)hipFree(p);
  mugiq_hip_rccl_comm_destroy(rccl);
  printf(rc == 0 ? "LOOP TEST PASSED\n" : "LOOP TEST FAILED\n");
  return rc;
}
