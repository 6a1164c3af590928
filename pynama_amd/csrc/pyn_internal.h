// Internal declarations shared by the translation units of libpynama_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <cstdint>
#include <cstdio>
#include <functional>
#include <string>
#include <vector>

#include "../../include/pynama_hip.h"

void pyn_set_error(const char* fmt, ...);

#define PYN_HIP(call)                                                                         \
  do {                                                                                        \
    hipError_t e_ = (call);                                                                   \
    if (e_ != hipSuccess) {                                                                   \
      pyn_set_error("%s:%d %s -> %s", __FILE__, __LINE__, #call, hipGetErrorString(e_));      \
      return PYN_EHIP;                                                                        \
    }                                                                                         \
  } while (0)

#define PYN_NCCL(call)                                                                        \
  do {                                                                                        \
    ncclResult_t r_ = (call);                                                                 \
    if (r_ != ncclSuccess) {                                                                  \
      pyn_set_error("%s:%d %s -> %s", __FILE__, __LINE__, #call, ncclGetErrorString(r_));     \
      return PYN_ENCCL;                                                                       \
    }                                                                                         \
  } while (0)

#define PYN_CHECK(cond, ...)                                                                  \
  do {                                                                                        \
    if (!(cond)) {                                                                            \
      pyn_set_error(__VA_ARGS__);                                                             \
      return PYN_EINVAL;                                                                      \
    }                                                                                         \
  } while (0)

#define PYN_TRY(call)                                                                         \
  do {                                                                                        \
    int rc_ = (call);                                                                         \
    if (rc_ != PYN_OK) return rc_;                                                            \
  } while (0)

// scratch device allocation released on every exit path of a set-up routine
struct DevTmp {
  void* p = nullptr;
  DevTmp() = default;
  DevTmp(const DevTmp&) = delete;
  DevTmp& operator=(const DevTmp&) = delete;
  ~DevTmp() {
    if (p) (void)hipFree(p);
  }
  hipError_t alloc(size_t bytes) {
    if (p) (void)hipFree(p);
    p = nullptr;
    return bytes ? hipMalloc(&p, bytes) : hipSuccess;
  }
  template <typename T>
  T* as() const {
    return static_cast<T*>(p);
  }
};

struct QuadTab {
  int ngp = 0;
  double* w = nullptr;       // [ngp]
  double* H = nullptr;       // [ngp][nn]
  double* Hrs = nullptr;     // [ngp][dim][nn]
  double* HrsCoo = nullptr;  // [ngp][dim][nc]
  // host facts about the tables (set by pyn_elem_tables_set)
  double wsum = 0.0;         // sum of the weights
  bool const_grad = false;   // Hrs and HrsCoo identical at every point and to each other (affine simplex)
};

struct DMat {
  int br = 0, bc = 0;
  double* val = nullptr;  // [nnzb*br*bc], layout in pynama_hip.h
  double* sell_val = nullptr;  // SELL-64 image of `val` (scalar matrices, solver side)
  bool sell_valid = false;     // the image holds the current values
  bool prod_ready = false;     // pyn_sell_ensure has chosen the product kernel for the current values (image / CSR values)
  bool csr_product = false;    // scalar dictionary-mode matrix: the product reads `val` directly (csrl_spmv_kernel), there is no image (decided in pyn_sell_ensure)
  bool csrlb_product = false;  // 2x2-block dictionary-mode matrix: lane per scalar row over LDS-staged runs of the block-CSR values (csrlb_spmv_kernel), no image
  bool bcsr_product = false;   // block matrix / long scalar rows: the product reads the block-CSR values directly (bcsr_spmv_kernel), no image either
  double* dinv = nullptr;      // 1 / diagonal per scalar row (Jacobi), written by the lattice assemblies in their store
  bool dinv_valid = false;     // phase, else extracted once per matrix version (pyn_dinv_ensure)
  double* lu = nullptr;        // dense LU factors of small systems (pyn_direct.hip), [n][n] row-major, multipliers in place
  int* lu_piv = nullptr;       // [n] row interchanges + [1] singular-column flag + [n] the same as a gather
  int64_t lu_n = 0;            // order the two buffers were sized for
  bool lu_valid = false;
  // "imposed-column" matrices (Krhs / Arhs of an assembly) are zero except in rows next to imposed nodes.  rhs_clean records for
  // which Dirichlet set (pyn_ctx::bc_stamp) the stored values are known to be exactly that matrix -- or all zero (PYN_RHS_ANY: a fresh
  // or zeroed matrix fits every set); the lattice kernels then leave the zero blocks of tiles without imposed nodes unwritten.
  int64_t rhs_clean = -2;      // PYN_RHS_UNKNOWN
  // COMPACT imposed-column matrix (pyn_mat_create_rhs): only the node rows with an imposed node in their neighbourhood (themselves
  // included) are stored -- what the reference preallocates for Krhs (src/matrices/mat_generator.py:42-58, 91: `drhs_nnz`).  Stored rows
  // keep the graph's full column list; `val` holds c_nnzb blocks.  The selection belongs to ONE Dirichlet set (c_stamp).
  bool rhs_compact = false;
  int64_t c_stamp = -1;        // pyn_ctx::bc_stamp of the selection (-1: none yet)
  int64_t c_nr = 0, c_nnzb = 0;
  int32_t* c_crow = nullptr;   // [n_owned] first block of the node's row in `val`, -1: row not stored
  int32_t* c_rsel = nullptr;   // [c_nr] stored node rows, ascending
  int32_t* c_cptr = nullptr;   // [c_nr + 1] first block of every stored row
  bool live = false;
  void touch() {               // the values are about to change
    sell_valid = prod_ready = dinv_valid = lu_valid = false;
    rhs_clean = -2;
  }
  void release_lu() {
    (void)hipFree(lu);
    (void)hipFree(lu_piv);
    lu = nullptr;
    lu_piv = nullptr;
    lu_n = 0;
    lu_valid = false;
  }
};

struct PatchPlan {
  bool user = false;  // set through the C ABI (not the automatic consecutive-row plan)
  int32_t *rowptr = nullptr, *rows = nullptr, *eptr = nullptr, *elem = nullptr;
  void *rowslot4 = nullptr, *kmap4 = nullptr;
  int npatch = 0, maxrows = 0, maxlen = 0;
  int64_t npe = 0;
};

// Structured-topology descriptor of a Q1 hex mesh (detected in pyn_mesh_set, verified entry by entry):
// nodes form nx*ny planes stacked in z; plane j (z order) starts at node id P[j]; element
// e = ix + (nx-1)*(iy + (ny-1)*l) sits between planes l and l+1.  Owned rows = planes
// [p_own0, p_own0 + n_own) (ids 0 .. n_owned-1); ghost planes of a rank's slab have ids >= n_owned.
struct Lattice {
  bool valid = false;
  int nx = 0, ny = 0, npl = 0, p_own0 = 0, n_own = 0;
  bool std_shape = false;      // all planes owned, in id order (single rank)
  int std_ok = -1;             // closed-form row offsets verified against the graph (-1: not checked yet)
  int32_t* d_P = nullptr;      // [npl]
  int32_t* d_zord = nullptr;   // [npl]: count | (dz+1) codes of the z-neighbour planes sorted by node id
};

// Structured topology of a SECOND-ORDER mesh (ngl = 3: 9-node quadrilaterals / 27-node hexahedra in the reference's local order,
// src/elements/spectral.py:346-431), the order all of the reference's cases run (src/cases/*.yaml: `ngl: 3`).  Nodes sit on the GLL
// lattice of 2 E + 1 points per axis; the slowest axis (y in 2-D, z in 3-D) is cut into "planes" (x-lines in 2-D) whose first node
// ids are P[j], so that a rank's slab with its ghost planes fits the same arithmetic: id = P[c_slow] + c_y NX + c_x (3-D).
struct Ho3Lattice {
  bool valid = false;
  int dim = 0, ngl = 0;         // ngl 3 (second order) or 2 (first order: the row-run kernels serve 2-D Q1 cells and the operators there)
  int EX = 0, EY = 0, EZ = 0;   // local elements per axis (2-D: EY = local element rows, EZ = 0)
  int NX = 0, NY = 0;           // nodes per x-line; x-lines per plane (3-D)
  int npl = 0, p_own0 = 0, n_own = 0;   // planes of the local mesh, owned ones = [p_own0, p_own0 + n_own) with ids 0 .. n_owned-1
  int32_t* d_P = nullptr;       // [npl]
  std::vector<int32_t> P;
  int affine = -1;              // every element a parallelogram / parallelepiped? (-1: not checked yet)
  int diag = 0;                 // ... and axis-aligned (J diagonal): set with `affine`
  double* d_geom = nullptr;     // [n_elem][6 | 10]: J^-1 (row = physical axis) and det J, rewritten by every assembly
  uint8_t* d_nbits = nullptr;   // per local node: bit p = DOF p imposed (packed copy of d_bcmask)
  int64_t nbits_stamp = -1;     // pyn_ctx::bc_stamp the packed copy belongs to
  uint8_t* d_runflag = nullptr; // per owned node that starts a run: does the run's node box hold an imposed DOF?
  int64_t runflag_stamp = -1;   // ... for this Dirichlet set
  int runflag_R = 0;            // ... and this run length
};

struct SellShape {
  int br = 0, bc = 0, maxw = 0;
  int64_t ns = 0, total = 0;
  int64_t* ptr = nullptr;   // [ns+1] slice offsets
  int* w = nullptr;         // [ns] slice widths (entries per scalar row)
  int32_t* col = nullptr;   // explicit expanded columns (only when the dictionary does not apply)
  int64_t int_begin = -1, int_end = -1;  // slices without ghost columns, when they are one contiguous range
};

struct DVec {
  int bs = 0;
  double* d = nullptr;  // [(n_owned+n_ghost)*bs]
  bool live = false;
};

constexpr int64_t PYN_RHS_UNKNOWN = -2, PYN_RHS_ANY = -1;   // DMat::rhs_clean
constexpr int PYN_MAX_PARTIALS = 2048;  // grid cap of every reducing kernel

struct pyn_ctx {
  int device = 0;
  hipStream_t stream = nullptr;
  hipStream_t comm_stream = nullptr;            // halo exchange overlapped with the interior SpMV rows
  hipEvent_t ev0 = nullptr, ev1 = nullptr;
  hipEvent_t ev_vec = nullptr, ev_halo = nullptr;  // vector ready for the exchange / ghosts have arrived
  double timers[PYN_T_COUNT] = {0};

  // communicator
  int rank = 0, nranks = 1;
  ncclComm_t comm = nullptr;            // all-reduces (main stream)
  ncclComm_t comm_halo = nullptr;       // halo exchanges (split off `comm`): either stream, never shares a communicator with an all-reduce
  struct pyn_shm_comm* shm = nullptr;   // TEST transport (pyn_comm_init_shm): host-staged exchange through POSIX shared memory, so
                                        // that several ranks can share ONE GPU (RCCL refuses that); never used by bench / product runs
  bool detached = false;  // ranks declared without a transport: ghosts are supplied by the caller
  // halo plan
  int64_t n_owned = 0, n_ghost = 0;
  std::vector<int> neigh;
  std::vector<int64_t> send_ptr, recv_ptr;
  int32_t* d_send_idx = nullptr;
  double* d_send_buf = nullptr;  // [n_send * max_bs]
  int64_t n_send = 0;
  bool halo_set = false;

  // mesh
  int dim = 0, nn = 0, nc = 0, ngl = 0;
  int64_t n_elem = 0, n_node = 0;
  int32_t* d_conn = nullptr;
  double* d_xyz = nullptr;
  QuadTab quad[3];
  int mesh_affine = -1;          // every element a parallelepiped? (-1: not checked yet; reset by pyn_mesh_set)
  bool aff_rw_standard = false;  // ... and so is int N_a d N_b (affine Rw path)
  bool aff_standard = false;  // the uploaded tables are those of the trilinear hexahedron in closed form
  bool q1_red_standard = false;    // reduced rule = the centroid, weight 8, trilinear tables (lean KLE kernel)
  bool q1_gauss_standard = false;  // ... pointwise: 2x2x2 Gauss rule, unit weights (lean general-geometry kernels)
  double* d_aff = nullptr;  // Q1-hex affine tables: [6][36] reference matrices + [4][8] non-affine monomial signs + [3][3] S

  // boundary condition
  int bc_ndof = 0;
  // matrix-free operators (pyn_matfree_set), slot = PYN_MATFREE_*: the Dirichlet mask is SNAPSHOT at set time, so later
  // pyn_bc_set calls (operator assembly, other matrices) do not change the operator
  bool mf_set[3] = {false, false, false};
  uint8_t* mf_mask[3] = {nullptr, nullptr, nullptr};   // null = no imposed DOF
  double mf_alpha_d = 0.0, mf_alpha_w = 0.0;
  int64_t bc_stamp = 0;  // bumped by every pyn_bc_set
  uint8_t* d_bcmask = nullptr;

  // node graph
  int32_t* d_rowptr = nullptr;
  int32_t* d_colidx = nullptr;
  int64_t nnzb = 0;

  // patch plans of the tiled assemblies (pyn_assemble_tiled.hip): [0] scalar forms, [1] KLE (3x3 blocks)
  PatchPlan plan[2];
  bool plan_unfit[2] = {false, false};  // the automatic plan did not fit this graph (reset by pyn_csr_symbolic)
  Lattice lat;  // structured topology, if the mesh has one (plan-free assembly kernel)
  Ho3Lattice ho3;   // ... of a second-order (ngl = 3) mesh (pyn_assemble_ho3.hip)
  // reference matrices of the ngl = 3 element in tensor (lattice) order, from the uploaded tables (pyn_elem_tables_set):
  // Tf / Tr[r][s][a][b] = sum_g w Hrs_r[a] Hrs_s[b] (full / reduced rule), Uf / Ur[r][a][b] = sum_g w H[a] Hrs_r[b]
  double* d_ho3_tabs = nullptr;
  std::vector<double> ho3_t1d_host;    // 1-D factors M, D, S of the three rules; the records are their tensor products when ho3_tens_ok
  double* d_ho3_t1d = nullptr;
  bool ho3_tens_ok = false;
  std::vector<double> ho3_tabs_host;   // host image of the records (the three rules arrive in separate pyn_elem_tables_set calls)
  bool ho3_tabs_ok[3] = {false, false, false};   // full, reduced, nodal rule
  int ho3_tabs_nn = 0;

  // SELL-64 structures, one per block shape, + the node-level column-pattern dictionary (pyn_sell.hip)
  std::vector<SellShape> sell_shapes;
  int32_t* sell_pid = nullptr;   // per-node column-pattern id (dictionary mode), else null
  int32_t* sell_tab = nullptr;   // [npat][32] relative node offsets
  int sell_npat = 0;
  bool sell_dict_built = false;

  std::vector<DMat> mats;
  std::vector<DVec> vecs;

  // reduction / solver scratch
  double* d_part = nullptr;    // [8][PYN_MAX_PARTIALS]
  double* d_scal = nullptr;    // [64] device scalars
  int* d_flag = nullptr;       // [8]  device flags (done, iters, reason ...)
  double* h_scal = nullptr;    // pinned host mirror [64]
  int* h_flag = nullptr;       // pinned host mirror [8]
  // CG work vectors (length n_local*bs_max), reallocated on demand
  double* d_work = nullptr;
  size_t work_bytes = 0;
  std::vector<hipEvent_t> prof_ev;  // event pool for per-kernel timing
  // 1/diagonal target of the scalar assembly in flight (the K matrix's DMat::dinv) and whether a kernel filled it
  double* asm_dinv = nullptr;
  bool asm_dinv_written = false;
  bool asm_rhs_clean = false;   // the Krhs / Arhs target of the assembly in flight holds zeros wherever this Dirichlet set leaves zeros
  const int32_t* asm_rcrow = nullptr;      // ... is a COMPACT matrix: first block of every owned node row in it (-1: not stored), else null
  const int32_t* asm_rcrow_fs = nullptr;   // the same for Krhsfs of the no-slip split
  bool asm_krhs_pending = false;           // the kernel family that took K left the compact Krhs to the generic kernel (run_assembly)
  // elements with an imposed node (the only ones that feed an imposed-column matrix), per Dirichlet set
  int32_t* d_esel = nullptr;
  int64_t n_esel = 0, esel_stamp = -1;
  // general-geometry KLE: off-diagonal element Laplacians [28][ne] between the pre-pass and the tile kernel (grown on demand)
  double* d_kle_lel = nullptr;
  size_t kle_lel_bytes = 0;
  // element-local scratch for pyn_elem_local
  double* d_eloc = nullptr;
  size_t eloc_bytes = 0;
};

inline int64_t n_local(const pyn_ctx* c) { return c->n_owned + c->n_ghost; }

// ---- cross-TU helpers ---------------------------------------------------------------------
int pyn_ensure_work(pyn_ctx* c, size_t bytes);
int pyn_halo_exchange(pyn_ctx* c, double* x, int bs);  // fills ghost part of x (stream ordered)
int pyn_halo_exchange_on(pyn_ctx* c, double* x, int bs, hipStream_t st);
int pyn_check_mat(pyn_ctx* c, int id, const char* what);
int pyn_check_vec(pyn_ctx* c, int id, const char* what);
int pyn_reduce_host(pyn_ctx* c, int nslots, int nblocks, int op, double* out);  // partials -> host, allreduced
int pyn_spmv_raw(pyn_ctx* c, const DMat& A, const double* x, double* y);        // no halo exchange
int pyn_extract_diag_inv(pyn_ctx* c, const DMat& A, double* dinv, bool invert);
int pyn_dinv_ensure(pyn_ctx* c, DMat& A);   // A.dinv valid for the current values (one diag_kernel per matrix version at most)
int pyn_sell_ensure(pyn_ctx* c, DMat& A, bool solver = true);   // solver: the product will be repeated (Krylov loop), not a one-off pyn_spmv
bool pyn_sell_supported(const DMat& A);
int pyn_sell_spmv(pyn_ctx* c, const DMat& A, const double* x, double* y, bool dot, int* grid_out);
void pyn_sell_drop_structure(pyn_ctx* c);
const SellShape* pyn_sell_shape(pyn_ctx* c, const DMat& A);
int pyn_sell_spmv_range2(pyn_ctx* c, const DMat& A, const double* x, double* y, bool dot, int64_t a0, int64_t a1, int64_t b0,
                         int64_t b1, int poff, int max_grid, hipStream_t st, int* grid_out);
int pyn_sell_spmv_range(pyn_ctx* c, const DMat& A, const double* x, double* y, bool dot, int64_t s0, int64_t s1, int poff,
                        int max_grid, hipStream_t st, int* grid_out);
// compact imposed-column matrices (pyn_rhs.hip)
int pyn_rhs_ensure(pyn_ctx* c, DMat& M, bool relayout = false);   // row selection + storage (values zeroed when laid out); relayout: for the CURRENT Dirichlet set
void pyn_rhs_release(DMat& M);
int pyn_rhs_expand(pyn_ctx* c, const DMat& M, double* full);   // full-pattern copy of the values (zeros in the rows not stored)
int pyn_bc_elements(pyn_ctx* c);                        // c->d_esel / n_esel for the current Dirichlet set
int64_t pyn_mat_blocks(const pyn_ctx* c, const DMat& M);        // blocks stored by the matrix (graph entries, or the compact count)
// entry i of the local connectivity as the host sees it: the array handed to pyn_mesh_set, or the closed form of pyn_mesh_box
using ConnAt = std::function<int32_t(int64_t)>;
int pyn_lattice_detect(pyn_ctx* c, const ConnAt& at);  // pyn_assemble_lattice.hip
bool pyn_q1_affine_tables_standard(const double* aff);
int pyn_mesh_all_affine(pyn_ctx* c, int* out);                        // pyn_assemble_tiled.hip
// collectives behind one switch: RCCL (product) or the shared-memory test transport
inline bool pyn_has_comm(const pyn_ctx* c) { return c->comm != nullptr || c->shm != nullptr; }
int pyn_allreduce_dev(pyn_ctx* c, double* dbuf, int n, int op, hipStream_t st);   // op 0 sum, 1 max; in place, device buffer
int pyn_lattice_symbolic(pyn_ctx* c, bool* done);
int pyn_assemble_lattice(pyn_ctx* c, double* A, double* Arhs, bool* handled);   // pyn_assemble_lattice.hip
bool pyn_lattice_matfree_supported(const pyn_ctx* c);
int pyn_lattice_matfree_spmv(pyn_ctx* c, const double* x, double* y, bool dot, int* grid_out);  // matrix-free Laplacian
int pyn_lattice_matfree_kle_spmv(pyn_ctx* c, const double* x, double* y, bool dot, int* grid_out);  // matrix-free KLE stiffness
int pyn_lattice_matfree_part(pyn_ctx* c, int op, const double* x, double* y, bool dot, int zsel, int part_off, int max_grid, hipStream_t st,
                             int* grid_out);   // tiles without (zsel 1) / with (zsel 2) ghost planes: halo overlap
int pyn_assemble_kle_lattice(pyn_ctx* c, double alpha_d, double alpha_w, double* K, double* Krhs, double* Rw, bool* handled);
bool pyn_q1_mixed_tables_standard(const double* w, const double* H, const double* Hrs);
bool pyn_q1_gauss_tables_standard(const double* w, const double* H, const double* Hrs, const double* HrsCoo);   // pyn_assemble_march.hip
int pyn_assemble_lattice_march(pyn_ctx* c, void* lat_args, int tile);   // general geometry, z-marching (pyn_assemble_march.hip)
// second-order (ngl = 3) lattices (pyn_assemble_ho3.hip)
int pyn_ho3_detect(pyn_ctx* c, const ConnAt& at);
void pyn_ho3_release(pyn_ctx* c);
int pyn_ho3_tables(pyn_ctx* c, int which, int ngp, const double* w, const double* H, const double* Hrs);
int pyn_ho3_symbolic(pyn_ctx* c, bool* done);
int pyn_assemble_ho3_lattice(pyn_ctx* c, int form, double alpha_d, double alpha_w, double* K, double* Krhs, double* Rw, bool* handled);
int pyn_assemble_ho3_operator(pyn_ctx* c, int rule, int br, int bc, int nterms, const int32_t* terms, const double* coef, double* M, bool* handled);
