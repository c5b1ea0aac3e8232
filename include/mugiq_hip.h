/*
 * mugiq_hip.h -- C ABI of libmugiq_hip.so, the MI355X-native (HIP / gfx950) drop-in for the
 * disconnected-loop hot path of ckallidonis/mugiq.
 *
 * Every entry point names the reference interface it replaces (file:line, relative to the MuGiq tree).
 * The reference's operator API is a set of C++ templates over QUDA types
 * (lib/contract_wrappers.cu, declared at include/loop_mugiq.h:280-311 and include/displace.h:109-111);
 * QUDA types cannot cross a C ABI, so each function takes plain pointers plus the POD descriptors below,
 * which carry exactly what the reference's Arg structs read out of a ColorSpinorField / cudaGaugeField
 * (include/contract_util.cuh:69-194).  include/mugiq_hip_operators.hpp re-declares the reference's
 * template names on top of this ABI; INTEGRATION.md shows the QUDA-side adapter.
 *
 * Conventions
 *  - all data pointers are DEVICE pointers unless the parameter name ends in _h;
 *  - every function returns 0 on success and a non-zero MugiqHipStatus on failure; the message is
 *    available from mugiq_hip_last_error() (the reference aborts through errorQuda instead:
 *    lib/contract_wrappers.cu:100,138,185);
 *  - `stream` is a hipStream_t passed as void* (NULL = the null stream).  Calls are asynchronous on
 *    that stream; nothing in this library calls hipDeviceSynchronize on the data path;
 *  - the caller owns every buffer.  Loop buffers are ACCUMULATED into (+=) and must be zeroed by the
 *    caller, as in the reference (lib/loop_mugiq.cpp:138,476).
 */
#ifndef MUGIQ_HIP_H
#define MUGIQ_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MUGIQ_HIP_VERSION 100

typedef enum MugiqHipStatus_e {
  MUGIQ_HIP_SUCCESS = 0,
  MUGIQ_HIP_ERROR_INVALID_ARGUMENT = 1, /* precondition the reference checks with errorQuda */
  MUGIQ_HIP_ERROR_UNSUPPORTED = 2,
  MUGIQ_HIP_ERROR_HIP = 3,              /* a HIP runtime call failed (reference: checkCudaError) */
  MUGIQ_HIP_ERROR_NO_DEVICE = 4
} MugiqHipStatus;

/* Values are those of QudaPrecision / QudaFieldOrder so an adapter can pass them through. */
#define MUGIQ_HIP_SINGLE_PRECISION 4
#define MUGIQ_HIP_DOUBLE_PRECISION 8
#define MUGIQ_HIP_FLOAT2_FIELD_ORDER 2
#define MUGIQ_HIP_FLOAT4_FIELD_ORDER 4

/* include/enum_mugiq.h:72-85 */
#define MUGIQ_HIP_DISP_DIR_X 0
#define MUGIQ_HIP_DISP_DIR_Y 1
#define MUGIQ_HIP_DISP_DIR_Z 2
#define MUGIQ_HIP_DISP_DIR_T 3
#define MUGIQ_HIP_DISP_SIGN_MINUS 0
#define MUGIQ_HIP_DISP_SIGN_PLUS 1

/*
 * A nSpin=4, nColor=3 colour-spinor field in QUDA's native even-odd layout: what
 * colorspinor::FieldOrderCB<Float,4,3,1,order> (include/contract_util.cuh:20-21) addresses.
 * Complex-element index of component k = 3*spin + colour at (parity, x_cb):
 *   FLOAT2: parity*parity_offset + k*stride + x_cb
 *   FLOAT4: parity*parity_offset + ((k/2)*stride + x_cb)*2 + (k%2)
 * ghost[dim][0|1] are the backward / forward ghost zones filled by a halo exchange of depth 1
 * (reference: ColorSpinorField::exchangeGhost, lib/contract_wrappers.cu:166-169); each zone is itself
 * laid out like a field body with volumeCB = stride = faceCB(dim), parity_offset = 12*faceCB(dim),
 * indexed by QUDA's ghostFaceIndex.  They are read only for partitioned dims (commDim[dim] != 0).
 */
typedef struct MugiqHipSpinorField_s {
  void *data;            /* ColorSpinorField::V() */
  int precision;         /* 4 | 8 */
  int field_order;       /* 2 | 4 */
  int nParity;           /* SiteSubset(); the hot path requires 2 (lib/contract_wrappers.cu:100,185) */
  int volumeCB;          /* VolumeCB() */
  int stride;            /* Stride() = volumeCB + pad */
  int X[4];              /* full local lattice dims (x,y,z,t), all even */
  int64_t parity_offset; /* complex elements between the two parities = Bytes()/2/sizeof(complex) */
  void *ghost[4][2];     /* may be all NULL on a single domain */
} MugiqHipSpinorField;

/*
 * The border-extended gauge field Displace builds (lib/displace.cpp:104-134), in QUDA's native
 * FLOAT2 gauge order with 18 reals per link (gauge_mapper<Float,QUDA_RECONSTRUCT_NO>,
 * include/contract_util.cuh:23-24): complex index of element (row,col) of link (dir, x_cb, parity)
 *   parity*parity_offset + (dir*9 + row*3 + col)*stride + x_cb
 * with x_cb the even-odd index on the EXTENDED lattice dimEx = X + 2*R.
 */
typedef struct MugiqHipGaugeField_s {
  void *data;
  int precision;         /* 4 | 8 */
  int X[4];              /* interior (non-extended) local dims */
  int R[4];              /* border per dim; reference uses 2*commDimPartitioned (lib/displace.cpp:16) */
  int stride;            /* extended volumeCB + pad */
  int64_t parity_offset; /* complex elements between parities */
} MugiqHipGaugeField;

/* ---- housekeeping ------------------------------------------------------------------------------- */
int mugiq_hip_version(void);
const char *mugiq_hip_last_error(void);
/* number of visible HIP devices (0 if none); does not initialise a context */
int mugiq_hip_device_count(void);
/* Streams and re-entrancy.  Every entry point that launches kernels takes the HIP stream they run on.  The small device
 * tables a call uploads for its kernels (eigenvector pointer lists, 1/sigma, momenta) and the workspaces the library
 * allocates when the caller passes none are kept PER (device, stream): calls issued on different streams -- from one
 * thread or several, through free operators or through different MugiqHipLoop objects -- are independent; calls on
 * one stream are ordered by the stream.  (The reference's wrappers are single-stream and not re-entrant: per-call
 * cudaMalloc + cudaDeviceSynchronize, lib/contract_wrappers.cu:93-114.)  Two host threads must not issue calls on the
 * SAME stream at the same time.  mugiq_hip_release_stream drains `stream` and frees what the library holds for it; call
 * it before destroying a stream that was passed to this library (optional: the buffers are small and are reused if the
 * handle value comes back). */
int mugiq_hip_release_stream(void *stream);

/* Roofline calibration (measurement aid, not part of the reference): stream-read `bytes` of buf_d once with 16-B
 * loads per lane (plain or non-temporal) and write nothing.  Time it with events to get the achievable HBM read
 * bandwidth of the device at hand. */
int mugiq_hip_probe_read_bandwidth(const void *buf_d, size_t bytes, int nonTemporal, void *stream);

/* Test aid (not part of the reference): overwrite the LDS of every CU with NaN bit patterns.  LDS is not cleared between kernels;
 * a kernel that reads a cell it never wrote computes with whatever the previous kernel left there, which is usually finite.  With
 * this call in front such a read shows up in the result (tests/test_gpu_driver.py::test_kernels_do_not_read_unwritten_lds). */
int mugiq_hip_debug_poison_lds(void *stream);

/* ---- gamma tables ------------------------------------------------------------------------------- */
/* copyGammaCoeffStructToSymbol<Float>()  lib/contract_wrappers.cu:6-19
 * copyGammaMapStructToSymbol<Float>()    lib/contract_wrappers.cu:26-43
 * The tables are compile-time constants of the HIP kernels, so these only validate `precision`;
 * they are kept so Loop_Mugiq::copyGammaToConstMem (lib/loop_mugiq.cpp:162-167) maps 1:1. */
int mugiq_hip_copy_gamma_coeff_to_symbol(int precision);
int mugiq_hip_copy_gamma_map_to_symbol(int precision);
/* Host copies of the tables the kernels were compiled with (include/gamma.h:32-71,99-109):
 * row_value[16][4][2] (re,im), column_index[16][4], map_sign[16], map_index[16]. Any pointer may be NULL. */
int mugiq_hip_get_gamma_tables(double *row_value_h, int *column_index_h, double *map_sign_h, int *map_index_h);
/* GammaName(m)  include/gamma.h:11-20 ; NULL if m is out of range */
const char *mugiq_hip_gamma_name(int m);

/* ---- a1/a2  ultra-local or displaced loop contraction ------------------------------------------------ */
/* performLoopContraction<Float,order>(loopData_d, eVecL, eVecR, sigma)   lib/contract_wrappers.cu:88-115
 * kernel loopContract_kernel lib/mugiq_contract_kernels.cu:45-122:
 *   loopData[tid + V*iG] += (1/sigma) * vL^dag(x) G(iG) vR(x),  tid = x_cb + parity*volumeCB, iG in [0,16)
 * loopData_d: complex<Float>[16*V] of the fields' precision. */
int mugiq_hip_perform_loop_contraction(void *loopData_d, const MugiqHipSpinorField *eVecL,
                                       const MugiqHipSpinorField *eVecR, double sigma, void *stream);

/* The fast path (new): the eigenvector loop of Loop_Mugiq::computeCoarseLoop (lib/loop_mugiq.cpp:478-503)
 * folded into one launch: loopData += sum_{n<nVec} (1/sigma[n]) vL_n^dag G vR_n with the 16 accumulators
 * held in registers, so each eigenvector is read once and loopData is touched once.
 * eVecL_h/eVecR_h: host arrays of nVec descriptors (same geometry, precision, order); sigma_h: host doubles
 * (eVals_sigma, cast to Float as at lib/loop_mugiq.cpp:479).  eVecR_h may equal eVecL_h (ultra-local). */
int mugiq_hip_perform_loop_contraction_batched(void *loopData_d, const MugiqHipSpinorField *eVecL_h,
                                               const MugiqHipSpinorField *eVecR_h, const double *sigma_h,
                                               int nVec, void *stream);

/* Mixed precision (BASELINE.json configs[3]; beyond the reference, which is single-typed end to end): loopPrecision = 8
 * with fp32 eigenvectors keeps the storage in fp32 and does all arithmetic, the 16-gamma accumulation over the
 * eigenvectors and the loop buffer in fp64.  loopPrecision = 0 or = the fields' precision is the plain call. */
int mugiq_hip_perform_loop_contraction_batched_mixed(void *loopData_d, int loopPrecision,
                                                     const MugiqHipSpinorField *eVecL_h,
                                                     const MugiqHipSpinorField *eVecR_h, const double *sigma_h,
                                                     int nVec, void *stream);

/* ---- a4/a5  covariant displacement --------------------------------------------------------------------- */
/* performCovariantDisplacementVector<Float,order>(dst, src, gauge, dispDir, dispSign)
 * lib/contract_wrappers.cu:171-198, kernel lib/mugiq_displace_kernels.cu:156-185:
 *   dst(x) = U_d(x) src(x+d)  (sign +)   |   dst(x) = U_d^dag(x-d) src(x-d)  (sign -)
 * commDim[4] replaces QUDA's global comm_dim_partitioned() (include/contract_util.cuh:89): for a
 * partitioned dim the neighbour of an on-face site is read from src->ghost[dim][bnd].  The halo
 * exchange itself (exchangeGhostVec, :166-169) is the caller's job -- see mugiq_hip_pack_face. */
int mugiq_hip_perform_covariant_displacement_vector(const MugiqHipSpinorField *dst,
                                                    const MugiqHipSpinorField *src,
                                                    const MugiqHipGaugeField *gauge, int dispDir,
                                                    int dispSign, const int commDim[4], void *stream);

/* Pack the face of `src` a neighbour needs as its depth-1 ghost zone, in ghostFaceIndex order
 * (the send half of ColorSpinorField::exchangeGhost, lib/contract_wrappers.cu:166-169).
 *   high = 0: low face  (x[dim] = 0)        -> the backward neighbour's ghost[dim][1]
 *   high = 1: high face (x[dim] = X[dim]-1) -> the forward  neighbour's ghost[dim][0]
 * face_d: 2*12*faceCB complex elements of the field's precision, laid out as a ghost zone. */
int mugiq_hip_pack_face(void *face_d, const MugiqHipSpinorField *src, int dim, int high, void *stream);

/* Batched, multi-layer version for the fused path: for every eigenvector n < nVec and layer j < layers pack the
 * face x[dim] = j (high = 0) or x[dim] = X[dim]-1-j (high = 1) into
 * faces_d[n][j] = one ghost zone (2*12*faceCB complex), contiguous over (n, j) so one message carries them all. */
int mugiq_hip_pack_face_layers(void *faces_d, const MugiqHipSpinorField *eVecs_h, int nVec, int dim, int high,
                               int layers, void *stream);

/* ---- reflected displacement entries (new) ----------------------------------------------------------------------
 * The reference computes the "+mu" and "-mu" entries of a displacement independently (lib/loop_mugiq.cpp:478-500).
 * Because W_{-k}(x) = W_{+k}(x - k mu)^dagger, gamma matrices commute with colour matrices and sigma_n is real,
 *     L^-_{k,G}(x) = eta_G conj( L^+_{k,G}(x - k mu) ),   L^+_{k,G}(x) = eta_G conj( L^-_{k,G}(x + k mu) ),
 * eta_G = +-1 with G^dagger = eta_G G.  dstSlot_d / srcSlot_d: one loop slot each (16*V complex of `precision`, the
 * dataPos layout loopData[tid + V*iG]); dstDispSign = sign of the entry being derived; length = k.  If
 * commDim[dispDir] != 0, ghostLayers_d holds the k boundary layers of the neighbour's source slot as
 * mugiq_hip_pack_loop_layers writes them (dst "-": the backward neighbour's HIGH layers; dst "+": the forward
 * neighbour's LOW layers). */
int mugiq_hip_reflect_displaced_loop(void *dstSlot_d, const void *srcSlot_d, const void *ghostLayers_d, const int localL[4],
                                     int dispDir, int dstDispSign, int length, const int commDim[4], int precision,
                                     void *stream);

/* layers_d[((j*16 + iG)*2 + parity)*faceCB + ghostFaceIndex] = slot value at x[dim] = X[dim]-layers+j (high = 1) or
 * x[dim] = j (high = 0), j < layers: 32*layers*faceCB complex of `precision`. */
int mugiq_hip_pack_loop_layers(void *layers_d, const void *slot_d, const int localL[4], int dim, int high, int layers,
                               int precision, void *stream);

/* ---- fused displaced contraction (new; the fast form of lib/loop_mugiq.cpp:485-497) --------------------------- */
/* For one displacement entry (dispDir, dispSign) and the lengths kValues_h[0..nK):
 *   loopData_d[slot i][tid + V*iG] += sum_n (1/sigma_n) v_n^dag(x) G(iG) W_k(x) v_n(x +- k mu),  k = kValues_h[i]
 * where W_k is the path-ordered link product, handed over as pathLinkFields_h[i] = device pointer to the field
 * E_k = D^k E_0, E_0(x)(s,c) = delta_sc for s < 3 (a FLOAT2 spinor field with stride = volumeCB, pad 0, of the
 * eigenvectors' precision) -- i.e. the output of k applications of
 * mugiq_hip_perform_covariant_displacement_vector to E_0.  The displaced vectors are never materialised.
 * Slots are 16*V complex apart.  If commDim[dispDir] != 0, ghostLayers_d holds `layers` >= max k face layers
 * of every eigenvector received from the neighbour the displacement points to, laid out as
 * mugiq_hip_pack_face_layers writes them (sign +: the forward neighbour's LOW layers; sign -: the backward
 * neighbour's HIGH layers). */
int mugiq_hip_displaced_loop_contraction_fused(void *loopData_d, const MugiqHipSpinorField *eVecs_h,
                                               const double *sigma_h, int nVec, const void *const *pathLinkFields_h,
                                               const int *kValues_h, int nK, int dispDir, int dispSign,
                                               const int commDim[4], const void *ghostLayers_d, int layers,
                                               void *stream);

int mugiq_hip_displaced_loop_contraction_fused_mixed(void *loopData_d, int loopPrecision,
                                                     const MugiqHipSpinorField *eVecs_h, const double *sigma_h, int nVec,
                                                     const void *const *pathLinkFields_h, const int *kValues_h, int nK,
                                                     int dispDir, int dispSign, const int commDim[4],
                                                     const void *ghostLayers_d, int layers, void *stream);

/* The same restricted to a region, so the halo transfer can overlap the part that does not need it:
 *   MUGIQ_HIP_REGION_INTERIOR: sites whose shifted reads x +- k mu stay inside the local lattice (ghostLayers_d not read,
 *                              may still be in flight);   MUGIQ_HIP_REGION_BOUNDARY: the remaining sites;
 *   INTERIOR followed by BOUNDARY on the same loop slots == MUGIQ_HIP_REGION_ALL. */
#define MUGIQ_HIP_REGION_ALL 0
#define MUGIQ_HIP_REGION_INTERIOR 1
#define MUGIQ_HIP_REGION_BOUNDARY 2
/* OR-ed into `region`: the addressed sites of the slots are WRITTEN instead of accumulated into -- the caller vouches that
 * they hold nothing yet and saves the memset of the slots and the read half of the read-modify-write. */
#define MUGIQ_HIP_REGION_OVERWRITE 0x100
int mugiq_hip_displaced_loop_contraction_fused_region(void *loopData_d, int loopPrecision,
                                                      const MugiqHipSpinorField *eVecs_h, const double *sigma_h, int nVec,
                                                      const void *const *pathLinkFields_h, const int *kValues_h, int nK,
                                                      int dispDir, int dispSign, const int commDim[4],
                                                      const void *ghostLayers_d, int layers, int region, void *stream);
/* The same, and in the same pass over the eigenvectors the ULTRA-LOCAL loop (displacement 0: the loop of lib/loop_mugiq.cpp:499-503,
 * sum_n (1/sigma_n) v_n^dag G v_n) into ultraLocalSlot_d (16*V complex, same region / overwrite semantics as the displaced slots):
 * it rides along as one more slot (k = 0, W = 1) of the tiled kernel when that has room -- a free slot of its 12-wave forms, or
 * the fourth slot of the 16-wave form (fp64 FLOAT2 column tiles with three lengths) -- which saves the separate pass over all
 * eigenvectors.  *carried = 1 if the slot was produced, 0 if not (then nothing was written to it and the caller computes it
 * with mugiq_hip_perform_loop_contraction_batched).  The slot is produced for the whole lattice or not at all: with a
 * region other than MUGIQ_HIP_REGION_ALL it is never taken along (*carried = 0).  ultraLocalSlot_d = NULL: plain _region call. */
int mugiq_hip_displaced_loop_contraction_fused_carry(void *loopData_d, int loopPrecision,
                                                     const MugiqHipSpinorField *eVecs_h, const double *sigma_h, int nVec,
                                                     const void *const *pathLinkFields_h, const int *kValues_h, int nK,
                                                     int dispDir, int dispSign, const int commDim[4],
                                                     const void *ghostLayers_d, int layers, int region,
                                                     void *ultraLocalSlot_d, int *carried, void *stream);

/* ---- a8  Fourier phase matrix -------------------------------------------------------------------------- */
/* createPhaseMatrixGPU<Float>(phaseMatrix_d, momMatrix_h, locV3, Nmom, FTSign, localL, totalL)
 * lib/contract_wrappers.cu:50-77, kernel lib/mugiq_util_kernels.cu:3-35.  commCoord[4] replaces QUDA's
 * comm_coord() (include/contract_util.cuh:64).  momMatrix_h: int[Nmom][3], MOM_MATRIX_IDX(id,im)=id+3*im. */
int mugiq_hip_create_phase_matrix(void *phaseMatrix_d, const int *momMatrix_h, long long locV3, int Nmom,
                                  int FTSign, const int localL[4], const int totalL[4],
                                  const int commCoord[4], int precision, void *stream);

/* ---- a9  even-odd -> time-major reorder with the G -> g5 G map ------------------------------------------ */
/* convertIdxOrder_mapGamma<Float>(dataPosMP_d, dataPos_d, nData, nLoop, nParity, volumeCB, localL)
 * lib/contract_wrappers.cu:133-156, kernel lib/mugiq_util_kernels.cu:59-99 */
int mugiq_hip_convert_idx_order_map_gamma(void *dataPosMP_d, const void *dataPos_d, int nData, int nLoop,
                                          int nParity, int volumeCB, const int localL[4], int precision,
                                          void *stream);

/* ---- a10  momentum projection (the cublasZgemm/Cgemm of lib/loop_mugiq.cpp:363-378) ---------------------- */
/* dataMom_d[M x N] = dataPosMP_d[M x K] * phaseMatrix_d[K x N], column-major, M = locT*nData, K = locV3,
 * N = Nmom, alpha = 1, beta = 0.  workspace_d may be NULL (the library then allocates its own scratch);
 * mugiq_hip_momentum_projection_workspace returns the bytes it would like. */
size_t mugiq_hip_momentum_projection_workspace(int locT, int nData, long long locV3, int Nmom, int precision);
int mugiq_hip_momentum_projection(void *dataMom_d, const void *dataPosMP_d, const void *phaseMatrix_d,
                                  int locT, int nData, long long locV3, int Nmom, int precision,
                                  void *workspace_d, size_t workspace_bytes, void *stream);

/* The same projection without the dense phase matrix (new).  exp(i s 2 pi p.x/L) factorises over x, y, z, so the sum over
 * the local spatial volume is taken one direction at a time, for the distinct p_x, then the distinct (p_x, p_y), then
 * the momenta: A is read once and the work drops from K*Nmom to about K*(number of distinct p_x) complex multiply-adds
 * per row.  Takes what createPhaseMatrixGPU takes (momMatrix_h [Nmom][3], FTSign, localL, totalL, commCoord) instead of
 * its output; the phases are rounded like the reference's, one direction at a time.  Same result to rounding. */
size_t mugiq_hip_momentum_projection_separable_workspace(const int *momMatrix_h, int Nmom, const int localL[4], int locT,
                                                          int nData, int precision);
int mugiq_hip_momentum_projection_separable(void *dataMom_d, const void *dataPosMP_d, const int *momMatrix_h, int Nmom,
                                            int FTSign, const int localL[4], const int totalL[4], const int commCoord[4],
                                            int locT, int nData, int precision, void *workspace_d, size_t workspace_bytes,
                                            void *stream);

/* a9 + a10 in one call (new): dataMom_d[M x Nmom] from the even-odd position-space buffer dataPos_d ([nData][V], the input
 * of convertIdxOrder_mapGamma).  The reorder with the G -> g5 G map and the sum over x happen in one kernel, so the
 * reordered copy dataPosMP is neither written nor read; then the y and z steps of the separable projection.
 * Workspace as for mugiq_hip_momentum_projection_separable. */
int mugiq_hip_convert_and_project(void *dataMom_d, const void *dataPos_d, int nData, int nLoop, const int *momMatrix_h, int Nmom,
                                  int FTSign, const int localL[4], const int totalL[4], const int commCoord[4], int precision,
                                  void *workspace_d, size_t workspace_bytes, void *stream);

/* The same for a SUBSET of the loop slots (new): only the slots slots_h[0 .. nSlots) of dataPos_d ([nLoop][16][V]) are read,
 * and only their rows t + locT*(ig + 16*slot) of dataMom_d (the full [locT*16*nLoop x Nmom] array) are written; the rows of the
 * other slots keep what they held.  The OPT plan uses it to leave reflected slots out of the projection (see
 * mugiq_hip_reflect_momentum_space). */
int mugiq_hip_convert_and_project_slots(void *dataMom_d, const void *dataPos_d, int nLoop, const int *slots_h, int nSlots,
                                        const int *momMatrix_h, int Nmom, int FTSign, const int localL[4], const int totalL[4],
                                        const int commCoord[4], int precision, void *workspace_d, size_t workspace_bytes, void *stream);

/* Reflected displacement entries in momentum space (new; host arrays only, no device work).  With L^-_k(x) = eta conj(L^+_k(x - k mu))
 * (mugiq_hip_reflect_displaced_loop), the Fourier transform of the derived slot follows from that of its source slot:
 *   dst(p, ig, t) = eta(15-ig) exp(-+ i FTSign 2 pi p_mu k / totalL[mu]) conj( src(-p, ig, t) )      mu = x, y, z
 *   dst(p, ig, t) = eta(15-ig) conj( src(-p, ig, t +- k) )   (t periodic over totT)                   mu = t
 * upper signs for dstDispSign = "+" (derived from a "-" entry).  dataMom_bcast_h: the gathered array of
 * performMomentumProjection (lib/loop_mugiq.cpp:415-424: per time-rank slabs of t + locT*ig + locT*16*iL + locT*16*nLoop*im);
 * slot srcSlot must be complete, slot dstSlot is overwritten.  MUGIQ_HIP_ERROR_UNSUPPORTED if some momentum of the list has no
 * partner -p in it. */
int mugiq_hip_reflect_momentum_space(void *dataMom_bcast_h, int precision, int Nmom, const int *momMatrix_h, int FTSign,
                                     const int totalL[4], int nLoop, int locT, int totT, int dstSlot, int srcSlot, int dispDir,
                                     int dstDispSign, int length);

/* ==== f2: MG coarse path -- Loop_Mugiq::prolongateEvec (lib/loop_mugiq.cpp:277-319) = QUDA Transfer::P ============== */

/* A coarse-grid colour-spinor in QUDA's FLOAT2 order (what Eigsolve_Mugiq hands over when computeCoarse is set,
 * lib/loop_mugiq.cpp:482): nSpin = 4/spin_block_size = 2 chiralities, nColor = n_vec; complex index of (s, c) at
 * (parity, x_cb): parity*parity_offset + (s*nColor + c)*stride + x_cb, even-odd on the coarse lattice X. */
typedef struct MugiqHipCoarseField_s {
  void *data;
  int precision; /* 4 | 8 */
  int nSpin;     /* 2 */
  int nColor;    /* n_vec */
  int volumeCB;
  int stride;
  int X[4];      /* coarse lattice dims = fine dims / geo_block_size, all even */
  int64_t parity_offset;
} MugiqHipCoarseField;

/* One level of QUDA's Transfer: the block-orthonormal null vectors V as a field of the FINER side with a packed vector
 * index.  Finest level: FieldOrderCB<Float,4,3,n_vec,FLOAT2>, complex index of V(parity, x_cb; s, c, j) =
 * parity*parity_offset + ((3*s + c)*nVec + j)*stride + x_cb, spinBlockSize 2.  Coarse -> coarse levels: see
 * mugiq_hip_prolongate_coarse_batched.  (Producing V is QUDA's MG setup: out of scope.) */
typedef struct MugiqHipTransfer_s {
  const void *V;
  int precision;       /* 4 | 8 */
  int nVec;            /* mg_param.n_vec[0], default 24 (tests/loop.cpp:492) */
  int geoBlockSize[4]; /* mg_param.geo_block_size[0], default 4^4 (tests/loop.cpp:471) */
  int spinBlockSize;   /* 2 (tests/loop.cpp:569) */
  int X[4];            /* local dims of the finer side of this level */
  int stride;          /* volumeCB of the finer side + pad */
  int64_t parity_offset;
} MugiqHipTransfer;

/* fine_h[n](x; s, c) = sum_j V(x; s, c, j) * coarse_h[n](X(x); s/spinBlockSize, j) for all n < nVec in one launch
 * (the reference prolongs one eigenvector per call, and again for every displacement entry: lib/loop_mugiq.cpp:482). */
int mugiq_hip_prolongate_batched(const MugiqHipSpinorField *fine_h, const MugiqHipCoarseField *coarse_h, int nVec,
                                 const MugiqHipTransfer *transfer, void *stream);
/* One COARSE -> COARSE level of the hierarchy (mg_env.transfer[lev], lev >= 1; Loop_Mugiq::prolongateEvec walks them from
 * the coarsest level up before the finest transfer, lib/loop_mugiq.cpp:306-311).  Both sides are coarse fields with
 * nSpin = 2: out_h[n](x; s, c) = sum_j V(x; s, c, j) * in_h[n](X(x); s, j), with `transfer` describing this level:
 * X = dims of the FINER of the two lattices (= out_h[n].X), geoBlockSize, nVec = in_h[n].nColor, spinBlockSize = 1, and
 * V = FieldOrderCB<Float, 2, out_h[n].nColor, nVec, FLOAT2>: complex index parity*parity_offset +
 * ((out.nColor*s + c)*nVec + j)*stride + x_cb.  All nVec eigenvectors in one launch. */
int mugiq_hip_prolongate_coarse_batched(const MugiqHipCoarseField *out_h, const MugiqHipCoarseField *in_h, int nVec,
                                        const MugiqHipTransfer *transfer, void *stream);
/* Ultra-local loop of the MG path without materialising the fine vectors:
 * loopData += sum_n (1/sigma_n) (P c_n)^dag G (P c_n).  loopPrecision as in the *_mixed entry points. */
int mugiq_hip_prolongate_contract_batched(void *loopData_d, int loopPrecision, const MugiqHipCoarseField *coarse_h,
                                          const double *sigma_h, int nVec, const MugiqHipTransfer *transfer,
                                          void *stream);

/* ==== host-side driver: the Loop_Mugiq / Displace classes of the reference ========================================= */

/* include/enum_mugiq.h:35-41.  calcType is parsed but never read by the reference's live code; here it selects
 * the execution plan: BASIC = the reference's own sequence (one displacement + one contraction launch per
 * eigenvector and step, lib/loop_mugiq.cpp:478-503); OPT and BLAS = eigenvector-batched contraction and the
 * fused displaced contraction (same results to rounding). */
#define MUGIQ_HIP_LOOP_CALC_TYPE_BLAS 0
#define MUGIQ_HIP_LOOP_CALC_TYPE_OPT_KERNEL 1
#define MUGIQ_HIP_LOOP_CALC_TYPE_BASIC_KERNEL 2

/* What Loop_Mugiq reads from QUDA's comm layer and MPI (lib/loop_mugiq.cpp:61-88,406-424) and what
 * exchangeGhostVec does (lib/contract_wrappers.cu:166-169), as callbacks so the host program owns the
 * transport (MPI in a MuGiq build; torch.distributed/RCCL in mugiq_amd; NULL comm = one process).
 * All callbacks return 0 on success. */
typedef struct MugiqHipComm_s {
  void *ctx;
  int rank, size;
  int grid[4];  /* comm_dim(d): ranks along x,y,z,t */
  int coord[4]; /* comm_coord(d) */
  /* Send `bytes` from send_d to the neighbour at coord[dim]+dir (dir = +1 | -1, periodic) and receive `bytes`
   * into recv_d from the neighbour at coord[dim]-dir.  Device pointers; ordered after prior work on `stream`
   * and complete (or stream-ordered) before later work on `stream`. */
  int (*sendrecv)(void *ctx, const void *send_d, void *recv_d, size_t bytes, int dim, int dir, void *stream);
  /* MPI_Reduce(SUM) over the ranks sharing coord[3] onto the one with coord[0..2] == 0 (COMM_SPACE,
   * lib/loop_mugiq.cpp:67,406).  Host buffers of n_real reals of `precision` bytes each. */
  int (*reduce_space)(void *ctx, const void *send_h, void *recv_h, size_t n_real, int precision);
  /* MPI_Gather over the ranks with coord[0..2] == 0, ordered by coord[3], root coord[3] == 0 (COMM_TIME,
   * lib/loop_mugiq.cpp:81,420-422).  recv_h is significant on the root only. */
  int (*gather_time)(void *ctx, const void *send_h, void *recv_h, size_t n_real_per_rank, int precision);
  /* MPI_Bcast from world rank 0 (lib/loop_mugiq.cpp:424) */
  int (*bcast)(void *ctx, void *buf_h, size_t n_real, int precision);
  /* Optional (both NULL or both set).  The OPT plan posts the eigenvector halos of ALL partitioned entries at the start
   * of a compute, in blocks of eigenvectors (about 2 GiB per message): for every block it issues, between group_begin and
   * group_end, one sendrecv per such entry -- all on the same stream, to different neighbours -- and the groups of
   * successive blocks follow each other on that stream.  A transport that can run the messages of a group concurrently
   * (different xGMI links: ncclGroupStart/End; MPI_Isend/Irecv + Waitall) may defer them until group_end(ctx, stream); one
   * without these members runs every sendrecv as it comes. */
  int (*group_begin)(void *ctx);
  int (*group_end)(void *ctx, void *stream);
  /* comm_dim_partitioned(d) beyond grid[d] > 1 (QUDA: comm_dim_partitioned_set(d), the `--partition` switch of its tests):
   * non-zero on an axis of extent 1 runs the PARTITIONED code path along it -- ghost zones, face packing, halo messages,
   * interior / boundary split, gauge borders from sendrecv -- with the rank as its own forward and backward neighbour, so
   * the halo machinery can be exercised (and timed) at full per-GPU size on one device.  The result equals the
   * unpartitioned one.  sendrecv must then be set even when size == 1.  Zero-initialise for the usual behaviour. */
  int partitioned[4];
} MugiqHipComm;

/* ---- a transport inside the library: the table above served by RCCL (csrc/comm_rccl.cpp) ---------------------------------
 * ncclSend / ncclRecv on the caller's stream inside ncclGroupStart / End for the halos (different neighbours = different xGMI
 * links at once), ncclReduce / ncclAllGather / ncclBroadcast over sub-communicators of the world one (ncclCommSplit) for the
 * COMM_SPACE / COMM_TIME steps of lib/loop_mugiq.cpp:61-88, 406-424 (host payloads staged through device buffers).  librccl is
 * loaded on first use.  Rank <-> coordinate map: QUDA's default (x slowest, t fastest).  One process per GPU; the device the
 * communicator lives on is the current device at creation.
 *   rank 0:        mugiq_hip_rccl_get_unique_id(id)           then hand the 128 bytes to every rank (MPI_Bcast, a file, a socket)
 *   every rank:    mugiq_hip_rccl_comm_create(&rc, id, rank, size, grid, partitioned)     (collective: ncclCommInitRank + 2 splits)
 *                  mugiq_hip_rccl_comm_fill(rc, &comm)        `comm` then goes wherever a MugiqHipComm goes
 *                  ...                                        mugiq_hip_rccl_comm_destroy(rc) once no loop object uses `comm` any more
 * A host that has an ncclComm_t already wraps it with mugiq_hip_rccl_comm_from_nccl (the communicator stays the host's).
 * `partitioned` as MugiqHipComm.partitioned (NULL = none).  Verified on hardware with one rank only (self-neighbour halos through
 * ncclSend / ncclRecv, tests/test_gpu_nccl.py): the pool this was built on has one GPU per box. */
typedef struct MugiqHipRcclComm_s MugiqHipRcclComm;
int mugiq_hip_rccl_get_unique_id(void *id128_out);
int mugiq_hip_rccl_comm_create(MugiqHipRcclComm **out, const void *id128, int rank, int size, const int grid[4], const int partitioned[4]);
int mugiq_hip_rccl_comm_from_nccl(MugiqHipRcclComm **out, void *ncclComm_world, const int grid[4], const int partitioned[4]);
int mugiq_hip_rccl_comm_fill(MugiqHipRcclComm *c, MugiqHipComm *out);
int mugiq_hip_rccl_comm_destroy(MugiqHipRcclComm *c);
/* Multi-path halos (off by default; needs more than two ranks).  A halo message goes to ONE neighbour, i.e. over one of a GPU's
 * seven xGMI links, while the links to the GPUs that are no neighbour on that axis idle: with this on, every sendrecv of a transfer
 * group is cut into 1 + R parts (R <= 6 ranks that are neither the origin nor the destination); part 0 travels directly, the others
 * through one relay each, first hops in one ncclGroup, second hops in the next, through a bounce buffer the communicator owns (about
 * the size of the group's messages).  Every rank must switch it the same way.  The schedule (a pure function of rank, grid, axis,
 * direction and size) is exposed for inspection: mugiq_hip_rccl_relay_plan lists what `rank` posts for one message -- phase 1 | 2,
 * kind 0 send from the send buffer, 1 receive into the receive buffer, 2 receive into the bounce area, 3 send from the bounce area,
 * peer, byte offset and length -- and returns the number of operations (tests/test_rccl_relay_plan_cpu.py delivers every byte with
 * it on a simulated network).  Never run on hardware (one-GPU boxes). */
int mugiq_hip_rccl_comm_set_multipath(MugiqHipRcclComm *c, int on);
int mugiq_hip_rccl_relay_plan(int rank, const int grid[4], int dim, int dir, size_t bytes, int max_ops, int *phase, int *kind, int *peer,
                              size_t *offset, size_t *len, size_t *bounce_bytes);

/* exchangeGhostVec(ColorSpinorField *x), lib/contract_wrappers.cu:166-169 (x->exchangeGhost(QUDA_INVALID_PARITY, nFace = 1, 0)):
 * fill the depth-1 ghost zones v->ghost[d][0 | 1] of every partitioned dimension (comm->grid[d] > 1 or comm->partitioned[d]), both directions,
 * through comm->sendrecv (one transfer group when the transport has group_begin / group_end).  The zones must be device
 * buffers of 2*12*faceCB complex each; comm == NULL (one process) is a no-op.  The faces are packed into the library's
 * per-stream workspace. */
int mugiq_hip_exchange_ghost_vec(const MugiqHipSpinorField *v, const MugiqHipComm *comm, void *stream);

/* What Displace asks of QUDA's ColorSpinorField for its auxiliary vector (lib/displace.cpp:26-30: ColorSpinorField::Create
 * with QUDA_ZERO_FIELD_CREATE and setPrecision(coarsePrec_); :42,:50-51: operator=; :59: blas::zero), for hosts that do not
 * manage device memory themselves.  alloc: geometry, order, stride of `like`, `precision` (0 = like's), zeroed; ghost zones
 * (zeroed) for the dims with ghostDims[d] != 0 (NULL = none).  copy: same precision / order / geometry required. */
int mugiq_hip_alloc_spinor_like(MugiqHipSpinorField *out, const MugiqHipSpinorField *like, int precision, const int ghostDims[4]);
int mugiq_hip_free_spinor(MugiqHipSpinorField *f);
int mugiq_hip_copy_spinor(const MugiqHipSpinorField *dst, const MugiqHipSpinorField *src, void *stream);
int mugiq_hip_zero_spinor(const MugiqHipSpinorField *f, void *stream);

/* MugiqLoopParam (include/mugiq.h:28-47) with C arrays instead of std::vector/std::string.
 * gauge: the reference hands over host QDP-ordered links + a QudaGaugeParam and lets Displace build the
 * border-extended device field (lib/displace.cpp:104-134); here the extended device field is the input
 * (mugiq_amd builds it; see GaugeField).  May be NULL when doNonLocal == 0. */
typedef struct MugiqHipLoopParam_s {
  int Nmom;
  const int *momMatrix; /* [Nmom][3] */
  int FTSign;           /* LoopFTSign: -1 | +1 */
  int calcType;         /* MUGIQ_HIP_LOOP_CALC_TYPE_* */
  int writeMomSpaceHDF5;
  int writePosSpaceHDF5;
  int doMomProj;
  int doNonLocal;
  int nDispEntries;              /* disp_str.size() */
  const char *const *disp_entry; /* e.g. "+z:1,8" */
  const char *const *disp_str;   /* e.g. "+z" */
  const int *disp_start;
  const int *disp_stop;
  const char *fname_mom_h5;
  const char *fname_pos_h5;
  const MugiqHipGaugeField *gauge;
  int loopPrecision;             /* not in the reference: 0 = the eigenvectors' precision; 8 with fp32 eigenvectors = mixed
                                    precision (fp32 storage, fp64 accumulation, loop buffers, FT and output) */
} MugiqHipLoopParam;

/* Loop_Mugiq::LoopComputeParam + the element counts of allocateDataMemory
 * (include/loop_mugiq.h:141-271, lib/loop_mugiq.cpp:101-109) */
typedef struct MugiqHipLoopInfo_s {
  int nDispEntries, nLoop, nData, Nmom, precision, field_order, loopPrecision;
  int localL[4], totalL[4];
  int locT, totT;
  long long locV4, locV3, totV3;
  long long nElemPosLocPerLoop, nElemMomLocPerLoop, nElemMomTotPerLoop;
  long long nElemPosLoc, nElemMomLoc, nElemMomTot, nElemPhMat;
} MugiqHipLoopInfo;

typedef struct MugiqHipLoop_s MugiqHipLoop;

/* Loop_Mugiq::Loop_Mugiq(loopParams, eigsolve)  lib/loop_mugiq.cpp:6-59: takes what the class reads from
 * Eigsolve_Mugiq as a friend -- eVecs[0..nEv) and eVals_sigma[0..nEv) (lib/loop_mugiq.cpp:442,479) -- sets up
 * LoopComputeParam, allocates the data buffers, creates the phase matrix.  comm may be NULL (single process).
 * The descriptors are copied; the eigenvector memory stays the caller's. */
int mugiq_hip_loop_create(MugiqHipLoop **loop, const MugiqHipLoopParam *param, const MugiqHipSpinorField *eVecs_h,
                          const double *eVals_sigma_h, int nEv, const MugiqHipComm *comm, void *stream);
/* The same with eigsolve->useMGenv && eigsolve->computeCoarse (lib/loop_mugiq.cpp:42,482; configs[4]): the
 * eigenvectors are COARSE fields and are prolonged with `transfer` (mugiq_hip_prolongate_batched) -- once, not once
 * per displacement entry; without displacement entries the ultra-local loop runs through
 * mugiq_hip_prolongate_contract_batched and the fine vectors are never stored.  fineFieldOrder must be 2
 * ("Vector prolongation requires fieldOrder = FLOAT2", lib/loop_mugiq.cpp:283). */
int mugiq_hip_loop_create_coarse(MugiqHipLoop **loop, const MugiqHipLoopParam *param,
                                 const MugiqHipCoarseField *coarseEvecs_h, const double *eVals_sigma_h, int nEv,
                                 const MugiqHipTransfer *transfer, int fineFieldOrder, const MugiqHipComm *comm,
                                 void *stream);
/* The same for an MG hierarchy with nCoarseLevels = mg_param.n_level - 1 >= 1 coarse levels (include/mg_mugiq.h:20,30): the
 * eigenvectors live on the COARSEST level; transfers_h[0] is the finest transfer (fine lattice <-> level 1, spinBlockSize 2),
 * transfers_h[l], l >= 1, the one between level l and level l+1 (see mugiq_hip_prolongate_coarse_batched).  Every compute
 * prolongs all eigenvectors level by level (lib/loop_mugiq.cpp:306-314) into temporaries the loop object owns. */
int mugiq_hip_loop_create_coarse_levels(MugiqHipLoop **loop, const MugiqHipLoopParam *param,
                                        const MugiqHipCoarseField *coarsestEvecs_h, const double *eVals_sigma_h, int nEv,
                                        const MugiqHipTransfer *transfers_h, int nCoarseLevels, int fineFieldOrder,
                                        const MugiqHipComm *comm, void *stream);
/* Loop_Mugiq::computeCoarseLoop()  lib/loop_mugiq.cpp:439-525 (position-space loops for the ultra-local case and
 * every displacement entry, then performMomentumProjection :322-434 if doMomProj).  Synchronises `stream`. */
int mugiq_hip_loop_compute(MugiqHipLoop *loop);
int mugiq_hip_loop_get_info(const MugiqHipLoop *loop, MugiqHipLoopInfo *info);
/* slot bookkeeping of entry id: (dir, sign, start, stop, nLoopPerEntry, nLoopOffset) -> out6[6] */
int mugiq_hip_loop_get_entry(const MugiqHipLoop *loop, int id, int out6[6]);
/* After mugiq_hip_loop_compute: the entry that entry `id` was reflected from (see mugiq_hip_reflect_displaced_loop), or
 * -1 if it was computed from the eigenvectors; -2 for a bad handle / index. */
int mugiq_hip_loop_entry_derived_from(const MugiqHipLoop *loop, int id);
/* After mugiq_hip_loop_compute: the displacement entry whose pass over the eigenvectors also produced the ultra-local loop
 * (mugiq_hip_displaced_loop_contraction_fused_carry), or -1 if the ultra-local loop took a pass of its own. */
int mugiq_hip_loop_ultra_local_carrier(const MugiqHipLoop *loop);
/* After mugiq_hip_loop_compute: the number of posted halos whose face layers were written by the entry that runs first, on its way
 * through the eigenvectors, instead of by mugiq_hip_pack_face_layers beside it (fp64 FLOAT2, first entry along x on the row tile,
 * z / t partitioned; MUGIQ_HIP_PACK_IN_ENTRY=0 switches it off).  0: none; -1: bad handle / nothing computed yet. */
int mugiq_hip_loop_halos_packed_in_entry(const MugiqHipLoop *loop);
/* Phase timing of a compute (measurement aid; off by default).  When switched on, mugiq_hip_loop_compute brackets each
 * phase with a pair of HIP events on the stream the phase runs on and, after its final synchronisation, reports the
 * device time between them.  Phases of different streams overlap in time (that is the point of the halo stream). */
#define MUGIQ_HIP_PHASE_ULTRA_LOCAL 0          /* the ultra-local slot (lib/loop_mugiq.cpp:501-502 over all eigenvectors) */
#define MUGIQ_HIP_PHASE_ENTRY_FUSED 1          /* a displacement entry computed from the eigenvectors, one domain along its axis */
#define MUGIQ_HIP_PHASE_ENTRY_REFLECTED 2      /* an entry derived from its opposite-sign partner */
#define MUGIQ_HIP_PHASE_ENTRY_STEPWISE 3       /* an entry through the step-by-step sequence (BASIC plan, or length > local extent) */
#define MUGIQ_HIP_PHASE_MOMENTUM_PROJECTION 4  /* reorder + Fourier sums on the device */
#define MUGIQ_HIP_PHASE_HALO_TRANSFER 5        /* eigenvector halo on the halo stream; bytes = what this rank sends */
#define MUGIQ_HIP_PHASE_ENTRY_INTERIOR 6       /* partitioned entry: tiles that need no ghost layers */
#define MUGIQ_HIP_PHASE_ENTRY_BOUNDARY 7       /* partitioned entry: tiles that read the ghost layers */
#define MUGIQ_HIP_PHASE_PROLONGATION 8         /* MG path: coarse -> fine for all eigenvectors */
#define MUGIQ_HIP_PHASE_HALO_PREPARE 9         /* path-link fields + packing of the face layers of one entry; bytes packed */
#define MUGIQ_HIP_PHASE_HALO_WAIT 10           /* compute stream idle until the halo has landed (what the overlap did not hide) */
#define MUGIQ_HIP_PHASE_MOMENTUM_COPY 11       /* dataMom_d -> pinned host */
#define MUGIQ_HIP_PHASE_MOMENTUM_REDUCE 12     /* host: reduce over space ranks, gather over time ranks, broadcast (wall time) */
#define MUGIQ_HIP_PHASE_TOTAL_WALL 13          /* host wall time of the whole mugiq_hip_loop_compute call (always the last phase) */
#define MUGIQ_HIP_PHASE_MOMENTUM_REFLECT 15     /* host: reflected entries derived on the gathered momentum-space array (wall time) */
#define MUGIQ_HIP_PHASE_SCRATCH_ALLOC 14       /* host: hipMalloc of scratch / halo buffers the pool did not hold yet (wall time, bytes) */
typedef struct MugiqHipLoopPhase_s {
  int kind;     /* MUGIQ_HIP_PHASE_* */
  int entry;    /* displacement entry the phase belongs to, or -1 */
  double ms;    /* device time between the bracketing events (host wall time for kinds 12, 13) */
  double bytes; /* halo / copy phases: bytes moved by this rank; else 0 */
} MugiqHipLoopPhase;
int mugiq_hip_loop_set_profiling(MugiqHipLoop *loop, int on);
/* phases of the last compute, in issue order; returns their number (may exceed max_phases; out may be NULL to ask) */
int mugiq_hip_loop_get_phases(const MugiqHipLoop *loop, MugiqHipLoopPhase *out, int max_phases);
/* dataPos_d: [nLoop][16][V even-odd] complex, device.  dataPos (host) is copied on first request
 * (the reference copies it unconditionally at lib/loop_mugiq.cpp:512).
 * With momentum projection on, the OPT plan derives reflected displacement entries in MOMENTUM space
 * (mugiq_hip_reflect_momentum_space) and leaves their position-space slots out of the compute; the first call of either accessor
 * after such a compute produces them (mugiq_hip_reflect_displaced_loop), so what is returned is always complete.  On a process
 * grid with a reflected entry along a partitioned direction that first call exchanges the slots' boundary layers through
 * comm->sendrecv: every rank must make it.  MUGIQ_HIP_REFLECT_MOM=0 keeps everything in position space inside the compute. */
const void *mugiq_hip_loop_data_pos_d(const MugiqHipLoop *loop);
const void *mugiq_hip_loop_data_pos_h(MugiqHipLoop *loop);
/* dataMom_bcast (host): per time-rank slabs of t + locT*ig + locT*16*iL + locT*16*nLoop*im, concatenated in
 * coord[3] order (lib/loop_mugiq.cpp:415-424).  NULL before compute or without doMomProj. */
const void *mugiq_hip_loop_data_mom_bcast_h(const MugiqHipLoop *loop);
/* Loop_Mugiq::writeLoopsHDF5()  lib/loop_mugiq.cpp:668-693 -> writeLoopsHDF5_Mom :529-656: group tree
 * /mom_%+d_%+d_%+d/disp_0|disp_<+-dir>_<len>/<GammaName>/loop, dataset [totT][2] of native float|double.
 * World rank 0 writes the whole file with serial HDF5 from dataMom_bcast (libhdf5 is bound at run time; override the
 * library with MUGIQ_HIP_HDF5_LIB).  Position-space output is "Not supported yet!" as in the reference (:660-663). */
int mugiq_hip_loop_write_hdf5(MugiqHipLoop *loop);
/* The writer on its own (host only, no GPU needed): dataMom_bcast_h as laid out by performMomentumProjection
 * (lib/loop_mugiq.cpp:415-424); disp_start <= disp_stop already normalised; nLoop = 1 + sum(stop-start+1). */
int mugiq_hip_write_loops_hdf5_mom(const char *filename, const void *dataMom_bcast_h, int precision, int Nmom,
                                   const int *momMatrix, int nDispEntries, const char *const *disp_str,
                                   const int *disp_start, const int *disp_stop, int locT, int totT);
/* Loop_Mugiq::~Loop_Mugiq */
int mugiq_hip_loop_destroy(MugiqHipLoop *loop);

/* ---- a6 (setup): Displace::createExtendedCudaGaugeField  lib/displace.cpp:70-134 ------------------------------------ */
/* bytes of a pad-0 extended field: volExCB * 36 * 2 parities * sizeof(complex) */
size_t mugiq_hip_extended_gauge_bytes(const int X[4], const int R[4], int precision);
/* Allocate (zeroed, pad 0) and describe the extended field for local dims X and border R -- gParamEx of
 * lib/displace.cpp:104-124 -- for hosts that do not manage device memory themselves; release with
 * mugiq_hip_free_extended_gauge. */
int mugiq_hip_alloc_extended_gauge(MugiqHipGaugeField *gauge, const int X[4], const int R[4], int precision);
int mugiq_hip_free_extended_gauge(MugiqHipGaugeField *gauge);
/* Fill gauge->data (device, caller-allocated, descriptor complete) from the host links of the LOCAL lattice in
 * QDP order, qdpLinks_h[dir][(parity*V/2 + x_cb)*18 + (row*3+col)*2 + re/im] of cpuPrecision (4|8)
 * (loopParams.gauge[4], tests/loop.cpp:88,106,902-918), then fill the R-deep borders: neighbour slabs through
 * comm->sendrecv for partitioned dims (edges/corners included, like exchangeExtendedGhost), periodic wrap otherwise. */
int mugiq_hip_create_extended_gauge(const MugiqHipGaugeField *gauge, const void *const qdpLinks_h[4], int cpuPrecision,
                                    const MugiqHipComm *comm, void *stream);

/* ---- user syntax (tests/loop.cpp:607-705) -------------------------------------------------------------------------- */
/* Parse "+z:1,8;-x:3;+y:2,5".  Returns the number of entries (<= max_entries) or a negative MugiqHipStatus.
 * disp_str_out: max_entries x 4 chars ("+z\0"); start/stop as in setLoopParam. */
int mugiq_hip_parse_displace_entry_string(const char *entry_string, int max_entries, char *disp_str_out,
                                          int *disp_start_out, int *disp_stop_out);
/* Displace::WhichDisplaceFlag/Dir/Sign (lib/displace.cpp:137-202): "+x".."-t" -> dir, sign; non-zero if unparsable */
int mugiq_hip_parse_displacement(const char *disp_str, int *dir_out, int *sign_out);

#ifdef __cplusplus
}
#endif
#endif /* MUGIQ_HIP_H */
