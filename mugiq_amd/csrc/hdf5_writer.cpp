// f1: momentum-space loop writer producing the reference's HDF5 group tree
// (Loop_Mugiq::writeLoopsHDF5_Mom, lib/loop_mugiq.cpp:529-656; names include/gamma.h:11-20):
//   /mom_%+d_%+d_%+d / disp_0 | disp_<+-dir>_<len> / <GammaName(ig)> / loop   dataset [totT][2], native float|double
// The reference opens the file with parallel HDF5 on MPI_COMM_WORLD and lets every "time process" write its
// hyperslab at offset tCoord*locT from its local dataMom (:561,571,628-633).  Here world rank 0 writes the whole
// file with SERIAL HDF5 from dataMom_bcast, which holds the same numbers for all time slabs (:420-424); the file
// contents are identical.  libhdf5 is bound at run time (dlopen) so libmugiq_hip.so has no link-time dependency.
#include <dlfcn.h>

#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "internal.h"

namespace mugiq {

typedef int64_t hid_t;   // HDF5 >= 1.10
typedef int herr_t;
typedef unsigned long long hsize_t;

struct H5Api {
  void *handle = nullptr;
  herr_t (*H5open)(void) = nullptr;
  hid_t (*H5Fcreate)(const char *, unsigned, hid_t, hid_t) = nullptr;
  herr_t (*H5Fclose)(hid_t) = nullptr;
  hid_t (*H5Gcreate2)(hid_t, const char *, hid_t, hid_t, hid_t) = nullptr;
  herr_t (*H5Gclose)(hid_t) = nullptr;
  hid_t (*H5Screate_simple)(int, const hsize_t *, const hsize_t *) = nullptr;
  herr_t (*H5Sclose)(hid_t) = nullptr;
  hid_t (*H5Dcreate2)(hid_t, const char *, hid_t, hid_t, hid_t, hid_t, hid_t) = nullptr;
  herr_t (*H5Dwrite)(hid_t, hid_t, hid_t, hid_t, hid_t, const void *) = nullptr;
  herr_t (*H5Dclose)(hid_t) = nullptr;
  herr_t (*H5get_libversion)(unsigned *, unsigned *, unsigned *) = nullptr;
  hid_t *native_double = nullptr, *native_float = nullptr;
};

static int load_hdf5(H5Api &api) {
  const char *env = getenv("MUGIQ_HIP_HDF5_LIB");
  const char *cands[] = {env, "libhdf5.so", "libhdf5.so.103", "/opt/conda/lib/libhdf5.so", "libhdf5_serial.so", nullptr};
  for (int i = 0; i < 6 && !api.handle; i++)
    if (cands[i] && cands[i][0]) api.handle = dlopen(cands[i], RTLD_NOW | RTLD_LOCAL);
  if (!api.handle)
    return set_error(MUGIQ_HIP_ERROR_UNSUPPORTED, "writeLoopsHDF5: cannot load libhdf5 (set MUGIQ_HIP_HDF5_LIB): %s", dlerror());
#define H5SYM(name)                                                                                     \
  *reinterpret_cast<void **>(&api.name) = dlsym(api.handle, #name);                                     \
  if (!api.name) return set_error(MUGIQ_HIP_ERROR_UNSUPPORTED, "writeLoopsHDF5: libhdf5 lacks %s", #name);
  H5SYM(H5open) H5SYM(H5Fcreate) H5SYM(H5Fclose) H5SYM(H5Gcreate2) H5SYM(H5Gclose) H5SYM(H5Screate_simple) H5SYM(H5Sclose)
  H5SYM(H5Dcreate2) H5SYM(H5Dwrite) H5SYM(H5Dclose) H5SYM(H5get_libversion)
#undef H5SYM
  api.native_double = reinterpret_cast<hid_t *>(dlsym(api.handle, "H5T_NATIVE_DOUBLE_g"));
  api.native_float = reinterpret_cast<hid_t *>(dlsym(api.handle, "H5T_NATIVE_FLOAT_g"));
  if (!api.native_double || !api.native_float)
    return set_error(MUGIQ_HIP_ERROR_UNSUPPORTED, "writeLoopsHDF5: libhdf5 lacks the native type ids");
  unsigned maj = 0, min = 0, rel = 0;
  api.H5get_libversion(&maj, &min, &rel);
  if (maj != 1 || min < 10)
    return set_error(MUGIQ_HIP_ERROR_UNSUPPORTED, "writeLoopsHDF5: HDF5 %u.%u.%u found, need >= 1.10 (64-bit hid_t)", maj, min, rel);
  if (api.H5open() < 0) return set_error(MUGIQ_HIP_ERROR_UNSUPPORTED, "writeLoopsHDF5: H5open failed");
  return MUGIQ_HIP_SUCCESS;
}

// dataMom_bcast: [nTimeRanks][im][iL][ig][locT] complex of `precision` (lib/loop_mugiq.cpp:415-424)
int write_loops_hdf5_mom(const char *filename, const void *dataMom_bcast, int precision, int Nmom, const int *momMatrix,
                         int nDispEntries, const std::vector<std::string> &dispString, const std::vector<int> &dispStart,
                         const std::vector<int> &dispStop, int nLoop, int locT, int totT) {
  static H5Api api;
  int st;
  if (!api.handle && (st = load_hdf5(api))) return st;
  const hid_t H5P_DEFAULT_ = 0, H5S_ALL_ = 0;
  const unsigned H5F_ACC_TRUNC_ = 0x0002u;
  const hid_t dtype = precision == 8 ? *api.native_double : *api.native_float;
  const int nGamma = 16;
  const int nTimeRanks = totT / locT;
  const long long nElemMomLoc = (long long)nGamma * Nmom * locT * nLoop;
  const size_t real = (size_t)precision;

  hid_t file_id = api.H5Fcreate(filename, H5F_ACC_TRUNC_, H5P_DEFAULT_, H5P_DEFAULT_);  // :572
  if (file_id < 0)
    return set_error(MUGIQ_HIP_ERROR_INVALID_ARGUMENT, "writeLoopsHDF5_Mom: Cannot open filename %s. Check that directory exists!", filename);
  hsize_t tdims[2] = {(hsize_t)totT, 2};  // :564
  std::vector<unsigned char> line((size_t)totT * 2 * real);
  int rc = MUGIQ_HIP_SUCCESS;
  for (int im = 0; im < Nmom && !rc; im++) {
    char group1_tag[16];  // :579-586
    snprintf(group1_tag, sizeof(group1_tag), "mom_%+d_%+d_%+d", momMatrix[0 + 3 * im], momMatrix[1 + 3 * im], momMatrix[2 + 3 * im]);
    hid_t group1_id = api.H5Gcreate2(file_id, group1_tag, H5P_DEFAULT_, H5P_DEFAULT_, H5P_DEFAULT_);
    if (group1_id < 0) {
      rc = set_error(MUGIQ_HIP_ERROR_INVALID_ARGUMENT, "writeLoopsHDF5_Mom: cannot create group %s (duplicate momentum, or a tag truncated at 15 characters as in the reference)", group1_tag);
      break;
    }
    int iL = 0;
    for (int iDE = -1; iDE < nDispEntries && !rc; iDE++) {  // :589-597
      const int dStart = iDE == -1 ? 0 : dispStart[iDE], dStop = iDE == -1 ? 0 : dispStop[iDE];
      for (int idisp = dStart; idisp <= dStop && !rc; idisp++) {
        char group2_tag[10];  // :600-608 -- char[10]: lengths >= 10 are truncated exactly as in the reference
        if (iDE == -1) snprintf(group2_tag, sizeof(group2_tag), "disp_0");
        else snprintf(group2_tag, sizeof(group2_tag), "disp_%s_%d", dispString[iDE].c_str(), idisp);
        hid_t group2_id = api.H5Gcreate2(group1_id, group2_tag, H5P_DEFAULT_, H5P_DEFAULT_, H5P_DEFAULT_);
        if (group2_id < 0) {
          rc = set_error(MUGIQ_HIP_ERROR_INVALID_ARGUMENT,
                         "writeLoopsHDF5_Mom: cannot create group %s/%s (the reference's char[10] tag truncates displacement lengths >= 10, or an entry is repeated)",
                         group1_tag, group2_tag);
          break;
        }
        for (int ig = 0; ig < nGamma && !rc; ig++) {
          hid_t group3_id = api.H5Gcreate2(group2_id, mugiq_hip_gamma_name(ig), H5P_DEFAULT_, H5P_DEFAULT_, H5P_DEFAULT_);  // :613-617
          hid_t space = api.H5Screate_simple(2, tdims, nullptr);
          hid_t dset = api.H5Dcreate2(group3_id, "loop", dtype, space, H5P_DEFAULT_, H5P_DEFAULT_, H5P_DEFAULT_);  // :621
          // loopIdx = locT*ig + locT*nGamma*iL + locT*nGamma*nLoop*im, slab of time rank r at t offset r*locT   :561,628
          const long long loopIdx = (long long)locT * ig + (long long)locT * nGamma * iL + (long long)locT * nGamma * nLoop * im;
          for (int r = 0; r < nTimeRanks; r++)
            memcpy(line.data() + (size_t)r * locT * 2 * real,
                   static_cast<const unsigned char *>(dataMom_bcast) + ((size_t)r * nElemMomLoc + loopIdx) * 2 * real,
                   (size_t)locT * 2 * real);
          if (group3_id < 0 || space < 0 || dset < 0 || api.H5Dwrite(dset, dtype, H5S_ALL_, H5S_ALL_, H5P_DEFAULT_, line.data()) < 0)
            rc = set_error(MUGIQ_HIP_ERROR_HIP, "writeLoopsHDF5_Mom: Could not write data for (mom,disp,gamma) = (%d,%d,%d)", im, iL, ig);  // :634
          if (dset >= 0) api.H5Dclose(dset);
          if (space >= 0) api.H5Sclose(space);
          if (group3_id >= 0) api.H5Gclose(group3_id);
        }
        iL++;
        api.H5Gclose(group2_id);
      }
    }
    api.H5Gclose(group1_id);
  }
  api.H5Fclose(file_id);
  return rc;
}

}  // namespace mugiq
